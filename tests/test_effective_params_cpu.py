"""What a session reports back through x264_encoder_parameters() when it cannot run what was asked (host logic over the stand-in device, no GPU):
every one of x264's presets is either its stated toolset (config.c:1486-1498 names them through x264_param_default_preset) or the DOCUMENTED downgrade,
and each downgrade is said in the session's log (codec.c:1274-1283 routes pf_log to the driver's log window)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

X264_ANALYSE_I4x4, X264_ANALYSE_I8x8, X264_ANALYSE_PSUB16x16, X264_ANALYSE_PSUB8x8, X264_ANALYSE_BSUB16x16 = 0x0001, 0x0002, 0x0010, 0x0020, 0x0100
X264_RC_CQP, X264_RC_CRF, X264_RC_ABR = 0, 1, 2
ME = {"dia": 0, "hex": 1, "umh": 2, "esa": 3, "tesa": 4}

# x264's preset table (x264_param_apply_preset [x264-upstream] as host/param.cpp restates it; the reference reaches it at codec.c:1463):
# ref, me, subme, trellis, partitions (all = with p4x4)
PRESET = {
    "ultrafast": dict(refs=1, me="dia", subme=0, trellis=0, p4x4=False),
    "superfast": dict(refs=1, me="dia", subme=1, trellis=0, p4x4=False),
    "veryfast": dict(refs=1, me="hex", subme=2, trellis=0, p4x4=False),
    "faster": dict(refs=2, me="hex", subme=4, trellis=1, p4x4=False),
    "fast": dict(refs=2, me="hex", subme=6, trellis=1, p4x4=False),
    "medium": dict(refs=3, me="hex", subme=7, trellis=1, p4x4=False),
    "slow": dict(refs=5, me="hex", subme=8, trellis=2, p4x4=False),          # (x264 since 2018: slow keeps hex; --me umh starts at slower)
    "slower": dict(refs=8, me="umh", subme=9, trellis=2, p4x4=True),
    "veryslow": dict(refs=16, me="umh", subme=10, trellis=2, p4x4=True),
    "placebo": dict(refs=16, me="tesa", subme=11, trellis=2, p4x4=True),
}
# the documented downgrades of this path (INTEGRATION.md "what a session runs"; host/encoder.cpp x264_encoder_open): ref > 5 -> 5, subme > 9 -> 9,
# tesa -> esa (and RD refinement needs hex / umh: subme 7 there), p4x4 off, trellis needs CABAC + subme >= 6
def expected(name):
    e = dict(PRESET[name])
    notes = []
    if e["refs"] > 5: e["refs"] = 5; notes.append("ref ")
    if e["me"] == "tesa": e["me"] = "esa"; notes.append("tesa")
    if e["subme"] > 9: e["subme"] = 9; notes.append("subme")
    if e["subme"] >= 8 and e["me"] not in ("hex", "umh"): e["subme"] = 7; notes.append("subme")
    if e["trellis"] and e["subme"] < 6: e["trellis"] = 0; notes.append("trellis")
    if e["p4x4"]: e["p4x4"] = False; notes.append("p4x4")
    return e, notes


def _session(tmp_path, n, opts, w=64, h=48):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "stub")])
    out = str(tmp_path / "s.h264")
    r = subprocess.run([sys.executable, os.path.join(HERE, "stub", "run_host_b.py"), out, str(w), str(h), str(n), "5"] + opts, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("preset", list(PRESET))
def test_effective_parameters_are_the_preset_or_its_documented_downgrade(tmp_path, preset):
    info = _session(tmp_path, 2, ["preset=" + preset, "log=1", "qp=30", "keyint=4", "bframes=0", "scenecut=0", "rc-lookahead=0"])
    e, notes = expected(preset)
    assert info["refs"] == e["refs"], (preset, info["refs"])
    assert info["me"] == ME[e["me"]], (preset, info["me"])
    assert info["subme"] == e["subme"], (preset, info["subme"])
    assert info["trellis"] == e["trellis"], (preset, info["trellis"])
    assert bool(info["inter"] & X264_ANALYSE_PSUB8x8) == e["p4x4"], (preset, hex(info["inter"]))
    log = " | ".join(m for _, m in info["log"])
    for key in notes:                               # nothing is dropped silently
        assert key in log, (preset, key, log)
    if not notes:
        assert "not implemented" not in log, (preset, log)


def test_second_pass_without_statistics_support_keeps_its_rate_control(tmp_path):
    """VfW encoding type 4, pass N (vfw.cpp case 4: b_stat_read) in a session that cannot run 2-pass (no B pictures, no weightp 2: not on the DPB model)
    must run as single-pass ABR at i_bitrate, not at a constant quantiser (the driver's multipass setup is codec.c:1509-1533)"""
    st = str(tmp_path / "x.stats")
    open(st, "w").write("#options: none\n")
    info = _session(tmp_path, 3, ["log=1", "bitrate=300", "pass=2", "stats=" + st, "bframes=0", "weightp=0", "keyint=8", "scenecut=0", "rc-lookahead=0", "no-mbtree"])
    assert info["rc_method"] == X264_RC_ABR, info
    assert info["stat_read"] == 0 and info["stat_write"] == 0
    assert any("single pass" in m for _, m in info["log"])
    # ... and the first pass likewise
    info = _session(tmp_path, 3, ["log=1", "bitrate=300", "pass=1", "stats=" + st, "bframes=0", "weightp=0", "keyint=8", "scenecut=0", "rc-lookahead=0", "no-mbtree"])
    assert info["rc_method"] == X264_RC_ABR and info["stat_write"] == 0, info


def test_bench_refuses_a_library_override():
    """VERDICT r05 #5a: the benchmark measures the product's own libraries — with X264GPU_LIB / X264GPU_HOST_LIB in the environment (the hook the stub tests and
    tools/ use to put another library behind the ABI) bench.py exits non-zero before anything is measured."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for var in ("X264GPU_LIB", "X264GPU_HOST_LIB"):
        env = {k: v for k, v in os.environ.items() if k not in ("X264GPU_LIB", "X264GPU_HOST_LIB")}
        env[var] = "/nonexistent/libstandin.so"
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--cpu-frames", "0", "--e2e-frames", "0"], env=env, capture_output=True, timeout=120)
        assert r.returncode != 0 and var.encode() in r.stderr, (var, r.returncode, r.stderr[-300:])


def test_decoder_probe_reports_what_it_finds_without_a_gpu():
    """tools/decoder_probe.py: with no third-party decoder CLI on the box the probe says so and claims nothing (no encode, no GPU needed); the keys bench.py prints are there"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import decoder_probe
    found = decoder_probe.find_decoders()
    assert set(found) == {"cli", "libraries"}
    res = decoder_probe.probe()
    assert set(res) >= {"found", "decoder", "equal", "pictures", "seen_not_driven", "what"}
    if not found["cli"]:
        assert res["found"] is False and res["equal"] is None and res["decoder"] is None

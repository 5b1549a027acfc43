"""CPU: the session's DECISIONS — picture types (scenecut, --b-adapt 1, keyint / min-keyint, closed GOPs) and single-pass CRF quantisers (I / P from the
frame costs, B from its nearest references) — as host/encoder.cpp takes them on the stand-in device, against oracle/decide.py, a second restatement of
the same parts of libx264 written independently of the host code (numpy / plain python over the CPU checker's frame costs).  Reference consumers of
these decisions: codec.c:1786 (every ICM_COMPRESS), config.c:1504-1514 (CRF / ABR rate control of the dialog)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from x264vfw_amd.lib import Pic

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import decide as D  # noqa: E402


def host_session(tmp_path, w, h, n, seed, opts, offsets=False):
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "stub")])
    dump = tmp_path / "dump"
    dump.mkdir(exist_ok=True)
    env = dict(os.environ, X264GPU_DUMP_RECORDS=str(dump))
    r = subprocess.run([sys.executable, os.path.join(HERE, "stub", "run_host_b.py"), str(tmp_path / "s.h264"), str(w), str(h), str(n), str(seed)] + opts,
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    pics, offs = [], []
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    for k in range(n):
        raw = (dump / f"pic{k:04d}.bin").read_bytes()
        pics.append(Pic.from_buffer_copy(raw[:C.sizeof(Pic)]))
        offs.append(np.frombuffer(raw[-4 * nmb:], np.float32))          # the per-macroblock quantiser offsets handed to the device (zeros: none)
    return (info, pics, offs) if offsets else (info, pics)


@pytest.mark.parametrize("w,h,n,seed,scene,kind,opts,kw", [
    (176, 144, 40, 3, 13, "moving", ["crf=24", "keyint=30"], dict(crf=24.0, keyint=30)),                                   # scene cuts inside and beyond min-keyint
    (176, 144, 36, 5, 0, "static", ["crf=22", "keyint=16", "b-adapt=1"], dict(crf=22.0, keyint=16)),                       # static content: runs of three B pictures, the keyint limit
    (208, 112, 30, 7, 11, "moving", ["crf=26", "keyint=25", "b-adapt=0", "bframes=2"], dict(crf=26.0, keyint=25, b_adapt=0, bframes=2)),
    (176, 144, 28, 9, 9, "moving", ["crf=23", "keyint=40", "bframes=0", "weightp=2", "ref=3"], dict(crf=23.0, keyint=40, bframes=0)),   # no B pictures: P and scene cuts only
    (176, 144, 34, 13, 12, "moving", ["crf=24", "keyint=40", "b-adapt=2"], dict(crf=24.0, keyint=40, b_adapt=2)),                  # presets slower and up: the Viterbi search over the window
    (176, 144, 30, 15, 0, "static", ["crf=23", "keyint=30", "b-adapt=2", "bframes=5", "b-pyramid=none"], dict(crf=23.0, keyint=30, b_adapt=2, bframes=5, b_pyramid=0)),
    (176, 144, 32, 11, 0, "static", ["crf=25", "keyint=250", "b-pyramid=none", "ipratio=1.6", "pbratio=1.5", "qcomp=0.7"],
     dict(crf=25.0, keyint=250, b_pyramid=0, ip_factor=1.6, pb_factor=1.5, qcomp=0.7)),
])
def test_decisions_equal_the_twin(tmp_path, w, h, n, seed, scene, kind, opts, kw):
    sys.path.insert(0, os.path.join(HERE, "stub"))
    from run_host_b import make_frames
    common = ["no-mbtree", "aq-mode=0", "weightp=0" if not any(o.startswith("weightp") for o in opts) else "subme=7", "rc-lookahead=0"]
    info, pics = host_session(tmp_path, w, h, n, seed, opts + common + ([f"scene_len={scene}"] if scene else []) + (["static=1"] if kind == "static" else []))
    frames = make_frames(w, h, n, seed, scene_len=scene, static=int(kind == "static"))
    p = D.Params((w + 15) // 16, (h + 15) // 16, **kw)
    slots = 64
    st = O.OracleSlicetype(w, h, slots=slots, bframes=max(p.bframes, 1), me_method=info["me"], subme=info["subme"], me_range=info["me_range"], mv_range=info["mv_range"])
    twin = D.run_session(frames, p, st, slots)
    st.close()
    got = [(r[1], r[0]) for r in info["recs"]]                                   # (display index, x264 type) in coding order
    want = [(f, t) for f, t, _, _, _ in twin]
    assert got == want, f"picture types: session {got} twin {want}"
    assert len({t for _, t in want}) >= 2
    for k, ((f, t, qp, qpf, _off), pic) in enumerate(zip(twin, pics)):
        assert pic.qp == qp, f"coded picture {k} (display {f}, type {t}): quantiser {pic.qp} vs the twin's {qp} ({qpf:.3f})"
        assert pic.qpm == np.float32(qpf), (k, pic.qp, pic.qpm, qpf)          # the device gets x264's float quantiser (rc->qpm) as it is


DEFAULT_CASES = [
    # the driver's default session (codec.c:1504-1507, config.c:109-111): CRF + AQ mode 1 + macroblock-tree + b-adapt 1 + weightp 2, a short and the real rc-lookahead
    (176, 144, 40, 3, 14, "moving", 0, ["crf=23", "keyint=250", "rc-lookahead=10"], dict(crf=23.0, keyint=250, rc_lookahead=10, weightp=2)),
    (176, 144, 60, 5, 23, "moving", 0, ["crf=23", "keyint=250", "rc-lookahead=40"], dict(crf=23.0, keyint=250, rc_lookahead=40, weightp=2)),
    (208, 112, 36, 7, 0, "moving", 4, ["crf=21", "keyint=30", "rc-lookahead=12"], dict(crf=21.0, keyint=30, rc_lookahead=12, weightp=2)),          # a fade: lookahead weights; the keyint limit
    (176, 144, 30, 9, 11, "moving", 4, ["crf=24", "keyint=250", "rc-lookahead=8", "weightp=0"], dict(crf=24.0, keyint=250, rc_lookahead=8, weightp=0)),  # X264_WEIGHTP_FAKE: the fade's weightdelta
    (176, 144, 30, 11, 0, "static", 0, ["crf=25", "keyint=250", "rc-lookahead=6", "b-pyramid=none", "qcomp=0.7", "aq-strength=0.8"],
     dict(crf=25.0, keyint=250, rc_lookahead=6, weightp=2, b_pyramid=0, qcomp=0.7, aq=0.8)),
    # presets slower and up decide the B runs with the trellis (b-adapt 2) under the tree; one B picture a run; weightp 1 (fade weights without the duplicates); min-keyint keeps a scene cut an I picture
    (176, 144, 34, 13, 15, "moving", 0, ["crf=23", "keyint=250", "rc-lookahead=14", "b-adapt=2"], dict(crf=23.0, keyint=250, rc_lookahead=14, weightp=2, b_adapt=2)),
    (176, 144, 28, 15, 0, "moving", 4, ["crf=22", "keyint=250", "rc-lookahead=9", "bframes=1", "weightp=1"], dict(crf=22.0, keyint=250, rc_lookahead=9, weightp=1, bframes=1)),
    (208, 112, 30, 19, 9, "moving", 0, ["crf=26", "keyint=60", "min-keyint=20", "rc-lookahead=7", "ipratio=1.2", "pbratio=1.5"], dict(crf=26.0, keyint=60, min_keyint=20, rc_lookahead=7, weightp=2, ip_factor=1.2, pb_factor=1.5)),
]


@pytest.mark.parametrize("w,h,n,seed,scene,kind,fade,opts,kw", DEFAULT_CASES)
def test_default_session_equals_the_twin(tmp_path, w, h, n, seed, scene, kind, fade, opts, kw):
    """the DEFAULT session against the twin: picture types, integer and float quantisers and the per-macroblock quantiser offsets (AQ - macroblock-tree for the
    pictures kept as references, AQ for the others) of every coded picture, float for float"""
    sys.path.insert(0, os.path.join(HERE, "stub"))
    from run_host_b import make_frames
    info, pics, offs = host_session(tmp_path, w, h, n, seed, opts + ([f"scene_len={scene}"] if scene else []) + (["static=1"] if kind == "static" else []) + ([f"fade={fade}"] if fade else []), offsets=True)
    frames = make_frames(w, h, n, seed, scene_len=scene, static=int(kind == "static"), fade=fade)
    aqs = kw.pop("aq", 1.0)
    strength = float(np.float32(aqs) * np.float32(1.0397))
    p = D.Params((w + 15) // 16, (h + 15) // 16, mbtree=True, aq_strength=strength, subme=info["subme"], **kw)
    slots = 128
    # (the lookahead searches within the session's effective --mvrange: the level's limit at this picture size)
    st = O.OracleSlicetype(w, h, slots=slots, bframes=max(p.bframes, 1), me_method=info["me"], subme=info["subme"], me_range=info["me_range"], mv_range=info["mv_range"], do_edges=1)
    twin = D.run_session(frames, p, st, slots, aq_of=lambda f: O.aq_offsets(f, w, h, strength))
    st.close()
    got = [(r[1], r[0]) for r in info["recs"]]
    want = [(f, t) for f, t, _, _, _ in twin]
    assert got == want, f"picture types: session {got} twin {want}"
    assert {3, 5} <= {t for _, t in want}          # P and B pictures at least
    moved = 0
    for k, ((f, t, qp, qpf, off), pic) in enumerate(zip(twin, pics)):
        assert pic.qp == qp and pic.qpm == np.float32(qpf), f"coded picture {k} (display {f}, type {t}): quantiser {pic.qp} / {pic.qpm} vs the twin's {qp} / {qpf}"
        assert offs[k].tobytes() == np.asarray(off, np.float32).tobytes(), f"coded picture {k} (display {f}, type {t}): offsets differ at {np.nonzero(offs[k] != off)[0][:6]}"
        moved += int(t != 5 and (off != O.aq_offsets(frames[f], w, h, strength)).any())
    assert moved >= 3, "the tree moved nothing"
    # the explicit weights of every P picture (x264_weights_analyse of the picture about to be coded: scales / offsets around the guess, the chroma planes beside luma,
    # the unified chroma denominator) equal the twin's
    weighted = 0
    for (f, t, _qp, _qpf, _off), pic in zip(twin, pics):
        if t != 3 or not kw["weightp"] or f not in D.run_session.last_weights:
            continue
        wt, wl, wc = D.run_session.last_weights[f], pic.wl0[0], pic.wc0[0]
        if wt is None:
            assert not wl.on and not wc.on[0] and not wc.on[1], (f, wl.on, wl.scale, wl.denom, wl.offset)
            continue
        weighted += 1
        assert (wl.on, wl.scale, wl.denom, wl.offset) == (1,) + wt["luma"], (f, (wl.scale, wl.denom, wl.offset), wt)
        for ci in range(2):
            if wt["chroma"][ci] is None:
                assert not wc.on[ci], (f, ci, wt)
            else:
                assert (wc.on[ci], wc.scale[ci], wc.offset[ci], wc.denom) == (1,) + wt["chroma"][ci] + (wt["cdenom"],), (f, ci, (wc.scale[ci], wc.offset[ci], wc.denom), wt)
    if fade and kw["weightp"]:
        assert weighted >= 3, weighted
    if fade:          # the fade is seen by the lookahead's weight analysis; without --weightp it still enters the tree as the weightdelta (X264_WEIGHTP_FAKE)
        assert D.run_session.last_stats["lookahead_weights"] >= 3, D.run_session.last_stats
        assert (D.run_session.last_stats["weightdelta"] >= 2) == (kw["weightp"] == 0), D.run_session.last_stats


def test_zones_move_the_quantisers(tmp_path):
    """--zones (x264 get_zone / get_qscale / the constant-quantiser branch of x264_ratecontrol_start; reaches the driver through its extra command line, codec.c:831-999):
    under CRF the non-B pictures inside a zone take its quantiser (q=) or bitrate_factor times their bits (b=) — equal to the twin picture by picture, the B pictures
    following from their references; under constant quantiser a zone shifts every picture type by (its qp - the P quantiser) or by -6 log2f(factor)"""
    sys.path.insert(0, os.path.join(HERE, "stub"))
    from run_host_b import make_frames
    w, h, n, seed = 176, 144, 30, 21
    opts = ["crf=24", "keyint=250", "no-mbtree", "aq-mode=0", "weightp=0", "rc-lookahead=0", "zones=4,9,q=31/14,22,b=0.4/20,22,q=18"]
    info, pics = host_session(tmp_path, w, h, n, seed, opts)
    frames = make_frames(w, h, n, seed)
    p = D.Params((w + 15) // 16, (h + 15) // 16, crf=24.0, keyint=250, zones=[(4, 9, 'q', 31), (14, 22, 'b', 0.4), (20, 22, 'q', 18)])
    st = O.OracleSlicetype(w, h, slots=64, bframes=3, me_method=info["me"], subme=info["subme"], me_range=info["me_range"], mv_range=info["mv_range"])
    twin = D.run_session(frames, p, st, 64)
    st.close()
    assert [(r[1], r[0]) for r in info["recs"]] == [(f, t) for f, t, _, _, _ in twin]
    seen = {"q31": 0, "q18": 0, "b": 0, "out": 0}
    for (f, t, qp, qpf, _off), pic in zip(twin, pics):
        assert pic.qp == qp and pic.qpm == np.float32(qpf), (f, t, pic.qp, pic.qpm, qp, qpf)
        if t == 3:          # P pictures: the zone's own quantiser, or 0.4 times the bits = the quantiser scale / 0.4 = 7.9 quantiser steps up
            if 4 <= f <= 9: assert qp == 31; seen["q31"] += 1
            elif 20 <= f <= 22: assert qp == 18; seen["q18"] += 1
            elif 14 <= f <= 19: assert qp >= 32; seen["b"] += 1
            else: assert 25 <= qp <= 30; seen["out"] += 1
    assert all(seen.values()), seen
    # constant quantiser
    info, pics = host_session(tmp_path, w, h, 16, seed, ["qp=26", "keyint=250", "bframes=2", "b-adapt=0", "b-pyramid=none", "scenecut=0", "zones=3,8,q=20/9,12,b=2"])
    order = [r[1] for r in info["recs"]]
    for f, t, pic in zip(order, [r[0] for r in info["recs"]], pics):
        base = {1: 23, 3: 26, 5: 28}[t]          # ipratio 1.4 -> -3, pbratio 1.3 -> +2
        want = base + (20 - 26) if 3 <= f <= 8 else int(np.float32(base) - np.float32(6.0) * np.float32(np.log2(np.float32(2.0))) + np.float32(0.5)) if 9 <= f <= 12 else base
        assert pic.qp == want, (f, t, pic.qp, want)


@pytest.mark.parametrize("opts,kw", [
    (["bitrate=300", "keyint=40", "no-mbtree", "aq-mode=0", "weightp=0", "rc-lookahead=0"], dict(bitrate=300, keyint=40)),
    (["bitrate=150", "keyint=250", "rc-lookahead=10", "bframes=2", "qpstep=6", "ratetol=0.5"], dict(bitrate=150, keyint=250, bframes=2, rc_lookahead=10, weightp=2, mbtree=True, aq=1.0, qpstep=6, rate_tolerance=0.5)),
])
def test_single_pass_abr_equals_the_twin(tmp_path, opts, kw):
    """single-pass ABR (the driver's encoding type 3 and the first pass of type 4, codec.c:1509-1524): rate_estimate_qscale's 1-pass branch — the rate factor from the bits window, the
    overflow pull towards the target, the asymmetric qpstep clip — with x264_ratecontrol_end's feedback: the twin is fed the sizes the session's pictures really had and must arrive at
    the same picture types and float quantisers, picture by picture"""
    sys.path.insert(0, os.path.join(HERE, "stub"))
    from run_host_b import make_frames
    w, h, n, seed, scene = 176, 144, 48, 17, 19
    kw = dict(kw)
    info, pics = host_session(tmp_path, w, h, n, seed, opts + [f"scene_len={scene}"])
    frames = make_frames(w, h, n, seed, scene_len=scene)
    aqs = kw.pop("aq", 0.0)
    strength = float(np.float32(aqs) * np.float32(1.0397)) if aqs else 0.0
    p = D.Params((w + 15) // 16, (h + 15) // 16, aq_strength=strength, **kw)
    st = O.OracleSlicetype(w, h, slots=128, bframes=max(p.bframes, 1), me_method=info["me"], subme=info["subme"], me_range=info["me_range"], mv_range=info["mv_range"], do_edges=int(p.mbtree))
    twin = D.run_session(frames, p, st, 128, aq_of=(lambda f: O.aq_offsets(f, w, h, strength)) if strength else None, sizes=[r[4] for r in info["recs"]])
    st.close()
    assert [(r[1], r[0]) for r in info["recs"]] == [(f, t) for f, t, _, _, _ in twin]
    for k, ((f, t, qp, qpf, _off), pic) in enumerate(zip(twin, pics)):
        assert pic.qp == qp and pic.qpm == np.float32(qpf), f"coded picture {k} (display {f}, type {t}): quantiser {pic.qp} / {pic.qpm} vs the twin's {qp} / {qpf}"
    assert len({pic.qp for pic in pics}) >= 4          # the rate control moves


def test_second_pass_plan_equals_the_twin(tmp_path):
    """x264's init_pass2 (the driver's encoding type 4, codec.c:1516-1541) twice: the host's plan — every picture's quantiser scale and the bits expected before it — against
    oracle/decide.py init_pass2 over the statistics file the first pass wrote: equal to a part in 10^9 (the same double arithmetic in another language), on two bitrates"""
    w, h, n = 176, 144, 60
    st = str(tmp_path / "x264.stats")
    common = ["scene_len=23", "keyint=40", "bframes=3", "subme=5", "trellis=0", f"stats={st}"]
    host_session(tmp_path, w, h, n, 9, common + ["bitrate=260", "pass=1"])
    lines = open(st).read().splitlines()
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    for kbps in (260, 120):
        info, _pics = host_session(tmp_path, w, h, n, 9, common + [f"bitrate={kbps}", "pass=2", "plan=1"])
        plan = info["plan"]
        assert plan and plan["count"] == n
        E = D.init_pass2(lines, nmb, kbps)
        got_q, want_q = np.array(plan["new_qscale"]), np.array([e.new_qscale for e in E])
        assert np.allclose(got_q, want_q, rtol=1e-9, atol=0), (kbps, np.abs(got_q / want_q - 1).max())
        got_b, want_b = np.array(plan["expected_bits"]), np.array([e.expected_bits for e in E])
        assert np.allclose(got_b, want_b, rtol=1e-9, atol=1e-6), (kbps, np.abs(got_b - want_b).max())
        assert len({round(float(q), 6) for q in want_q}) > 10          # a curve, not a constant
        total = want_b.max() + max(D.qscale2bits(e, e.new_qscale) for e in E if e.expected_bits == want_b.max())
        assert abs(total / (kbps * 1000.0 * n / 25.0) - 1.0) < 0.01      # the plan adds up to the request
        # ... and the feedback on top of the plan (rate_estimate_qscale's 2-pass branch), fed the sizes this pass' pictures really had: the same quantisers picture by picture
        fb = D.pass2_quantisers(E, [r[4] for r in info["recs"]], kbps)
        assert [r[1] for r in info["recs"]] == [f for f, _, _ in fb]
        for k, ((f, qp, qpf), pic) in enumerate(zip(fb, _pics)):
            assert pic.qp == qp and pic.qpm == np.float32(qpf), (kbps, k, f, pic.qp, pic.qpm, qp, qpf)


def test_fade_weights_are_the_fades_ratio(tmp_path):
    """known answer for x264_weights_analyse as the host restates it: on a clip that fades to black by 5 % of the first picture's level a picture
    (luma scaled, chroma scaled towards 128) a P picture's explicit weight of reference 0 must be the ratio of the two pictures' fade levels —
    luma scale / 2^denom within 2 % and a small offset; chroma the same ratio with the offset that keeps 128 where it is"""
    w, h, n, fade = 176, 144, 13, 5
    info, pics = host_session(tmp_path, w, h, n, 3, ["qp=24", "keyint=60", f"fade={fade}", "bframes=3", "no-mbtree", "aq-mode=0", "weightp=2"])      # (medium's lookahead: b-adapt 1, scenecut)
    level = lambda i: max(0.0, 1.0 - i * fade / 100.0)
    checked = 0
    for k, pic in enumerate(pics):
        if pic.slice_type != 0 or not pic.nref[0]:          # P pictures (X264GPU_SLICE_P = 0)
            continue
        disp = info["recs"][k][1]
        # reference 0 of a P picture: the I / P picture coded last before it
        ref = [info["recs"][j][1] for j in range(k) if info["recs"][j][0] in (1, 2, 3)][-1]
        want = level(disp) / level(ref)
        wl = pic.wl0[0]
        assert wl.on, f"picture {disp}: no luma weight on a {100 * (1 - want):.0f} % darker picture"
        got = wl.scale / float(1 << wl.denom)
        assert abs(got - want) < 0.02 * want and abs(wl.offset) <= 2, (disp, got, want, wl.offset)
        wc = pic.wc0[0]
        for c in range(2):
            if wc.on[c]:
                gc = wc.scale[c] / float(1 << wc.denom)
                # (the chroma planes of this clip are nearly flat: the scale is a coarse estimate, the offset must still keep grey grey)
                assert abs(gc - want) < 0.08 * want and abs(wc.offset[c] - 128 * (1 - gc)) <= 3, (disp, c, gc, want, wc.offset[c])
        checked += 1
    assert checked >= 2


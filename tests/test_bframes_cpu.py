"""CPU: B pictures end to end on the checker side — oracle/analyse.c's B analysis (spatial direct, both lists' searches, bi-prediction with implicit
weights, B RD decision) -> records -> the product's host CABAC writer + DPB model (host/cabac.cpp, host/dpb.hpp: frame_num, POC type 0, list
modification, MMCO) -> the checker decoder (oracle/h264dec.cpp: list initialisation / modification / marking, B macroblock layer, spatial direct
and weighted prediction written from the standard) must reproduce the encoder's reconstruction of every picture.  The reference's consumers of B
pictures: output/matroska.c:201, codec.c:1822-1826."""
import numpy as np
import pytest

import bgop
import host_lib as HL
import oracle_lib as O
from synth import synth_frames

MEDIUM = dict(refs=3, dpb=4, weightb=1, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256,
              chroma_qp_offset=-2, trellis=63)


def run(w, h, types, seed, bframes=3, pyramid=1, weightp=0, pics_out=None, weights=None, frames=None, direct="spatial", **over):
    kw = dict(MEDIUM, **over)
    frames = frames if frames is not None else synth_frames(w, h, len(types), seed=seed)
    cfg = O.default_config(w, h, **kw)
    enc = O.OracleEncoder(cfg)
    stream, recons, order, pocs = bgop.encode_gop(HL, enc, frames, types, cfg, 20, 23, 25, kw["refs"], bframes, pyramid, weightp, pics_out, weights, direct)
    dec = O.h264_decode(stream, len(order), w, h)
    assert len(dec) == len(order)
    assert O.h264_last_pocs() == pocs
    for k, (d, r) in enumerate(zip(dec, recons)):
        assert np.array_equal(d, r), f"picture {k} (display {order[k][0]}, type {order[k][1]}) decodes differently"
    return stream, order


@pytest.mark.parametrize("w,h,types,seed,over", [
    (176, 144, "IBBBPBBBP", 5, {}),                                               # medium: bframes 3, b-pyramid, ref 3, weightb
    (96, 80, "IBPBBPBBBPP", 2, {}),                                               # runs of 1, 2, 3 B pictures and consecutive P pictures
    (208, 112, "IPBBBPBPBBP", 3, dict(weightb=0, mixed_refs=0)),
    (176, 144, "IBBBPBBBPBBBP", 6, dict(refs=1, dct8x8=0, trellis=0)),            # ref 1: the DPB still holds 4 pictures under b-pyramid
    (128, 96, "IBBPBBP", 7, dict(refs=5, dpb=5, chroma_me=0, psy_rd_q8=0)),
    (176, 144, "IBBPBP", 8, dict(dct_decimate=0)),
    (176, 144, "IBBBPBBP", 9, dict(me_method=2)),
    (128, 96, "IBPBBP", 12, dict(me_method=3, me_range=8)),
    (176, 144, "IBBBPBP", 13, dict(trellis=63 + 64)),
    (176, 288, "IBBBPBBP", 14, dict(slices=3)),                                   # x264 slice threads: B pictures in three slices (not filtered across)
    (176, 144, "IBBPBP", 15, dict(slices=9, slices_plain=1)),                     # --slices 9: one slice a macroblock row, filtered across
    (96, 160, "IBPBBP", 16, dict(slices=4, slices_plain=1, refs=2)),
    (176, 144, "IBBPBP", 19, dict(partitions=0x707)),                             # p8x8 without b8x8
    (176, 144, "IBBP", 20, dict(partitions=0xf06)),                               # b8x8 without p8x8
    # B analysis without RD (x264 below --subme 7: probe_bskip, the fast-skip search order, me_refine_qpel of the winner, SA8D/SATD transform choice)
    (176, 144, "IBBBPBBP", 21, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),      # preset fast's level minus one: subme 5
    (176, 144, "IBBPBP", 22, dict(rd=1, subme=6)),                                # --subme 6: RD in I/P slices, B slices one level down without it
    (128, 96, "IBBPBBP", 23, dict(rd=0, trellis=0, subme=2, psy_rd_q8=0, refs=1, partitions=0x303, mixed_refs=0, weightb=0)),      # veryfast: immediate skip on the probe
    (176, 144, "IBBBP", 24, dict(rd=0, trellis=0, subme=4, psy_rd_q8=0, refs=2, mixed_refs=0)),                                   # faster
    (96, 80, "IBPBBP", 25, dict(rd=0, trellis=0, subme=1, psy_rd_q8=0, refs=1, partitions=0x303, mixed_refs=0, weightb=0, dct8x8=0)),      # superfast
    (176, 144, "IBBP", 26, dict(rd=0, trellis=0, subme=3, psy_rd_q8=0, me_method=2)),
    # ... and with CAVLC (Main profile --no-cabac): mb_type ue(v) of Table 7-14, sub_mb_type, te(v) reference indices of both lists, B_Skip runs
    (176, 144, "IBBBPBBP", 41, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0, cabac=0)),
    (96, 80, "IBPBBPBBBPP", 42, dict(rd=0, trellis=0, subme=4, psy_rd_q8=0, cabac=0, refs=2)),
    (176, 144, "IBBPBP", 43, dict(rd=1, trellis=0, subme=6, cabac=0, partitions=0x707)),          # --subme 6: RD (CAVLC counts) in I / P slices only
    (176, 288, "IBBBPBBP", 44, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0, cabac=0, slices=3, dct8x8=0)),
    # ... and RD decisions of B slices on CAVLC bit counts (--no-cabac at subme 7 and up)
    (176, 144, "IBBBPBBP", 51, dict(cabac=0, trellis=0)),
    (96, 80, "IBPBBPBBBPP", 52, dict(cabac=0, trellis=0, refs=1, weightb=0, partitions=0xf07)),
    (176, 144, "IBBBPBBP", 18, dict(subme=9)),                                    # chroma-ME in B slices
    (176, 144, "IBBPBP", 19, dict(subme=9, rd=1 | 64)),                           # + deblock-aware RD
    # --subme 9 in full (i_mbrd 2 in B slices): intra_rd_refine, x264_me_refine_qpel_rd per list, x264_me_refine_bidir_rd of the bi-predicted parts
    (176, 144, "IBBBPBBP", 33, dict(subme=9, rd=63 | 64)),
    (96, 80, "IBPBBPBBBPP", 34, dict(subme=9, rd=63, partitions=0xf07)),
    (176, 144, "IBBPBP", 35, dict(subme=9, rd=63 | 64, trellis=127, me_method=2)),
    (128, 96, "IBBPBBP", 36, dict(subme=9, rd=3, refs=2, weightb=0)),               # the inter sites only
    (176, 144, "IBBBPBBP", 31, dict(subme=6)),                                    # preset fast: subme 6 + trellis 1
    (176, 144, "IBBPBP", 32, dict(subme=6, trellis=127)),
])
def test_b_pictures_decode_to_the_encoders_reconstruction(w, h, types, seed, over):
    run(w, h, types, seed, **over)


@pytest.mark.parametrize("w,h,types,seed,direct,over", [
    (176, 144, "IBBBPBBBP", 5, "temporal", {}),                                   # medium's structure: co-located pictures are P and B-ref pictures
    (96, 80, "IBPBBPBBBPP", 2, "temporal", {}),
    (176, 144, "IBBPBBP", 7, "temporal", dict(refs=5, dpb=5)),
    (208, 112, "IBBBPBBP", 3, "temporal", dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),          # without RD: probe_bskip on the temporal prediction
    (176, 144, "IBBBPBP", 9, "temporal", dict(me_method=2, partitions=0x707)),
    (176, 144, "IBBBPBBBPBBP", 5, "auto", {}),                                    # --direct auto: both modes probed, the running score picks the next picture's
    (176, 144, "IBBPBBPBBP", 8, "auto", dict(rd=0, trellis=0, subme=4, psy_rd_q8=0)),
    (128, 96, "IBBBPBBBP", 11, "auto", dict(subme=8, rd=63, me_method=1)),
])
def test_temporal_direct_and_direct_auto_decode(w, h, types, seed, direct, over):
    """x264 --direct temporal (mb_predict_mv_direct16x16_temporal: the co-located block's reference mapped into list 0, its vector scaled by the POC
    distances; no direct prediction where the co-located block has no list-0 motion) and --direct auto (each macroblock probes the skip under both
    modes, h->stat.i_direct_score picks the next B picture's mode): direct_spatial_mv_pred_flag, the checker decoder's 8.4.1.2.3"""
    pics = []
    run(w, h, types, seed, pics_out=pics, direct=direct, **over)
    if direct == "auto":
        marks = [p for p in pics if p[0] == "direct"]
        assert marks and all(sum(m[2]) > 0 for m in marks), marks            # both modes were probed and something could be skipped


@pytest.mark.parametrize("w,h,types,seed,over", [
    (176, 144, "IPPPPP", 4, {}),                                                  # P pictures only: lists of 1, then ref0 + duplicate + ...
    (176, 144, "IBBBPBBBPBP", 5, {}),                                             # medium
    (128, 96, "IPPBBPPP", 8, dict(refs=5, dpb=5, mixed_refs=0)),
    (96, 80, "IPPPP", 2, dict(refs=2, rd=0, trellis=0, subme=5, psy=0, psy_rd_q8=0)),
    (208, 112, "IPPPP", 3, dict(me_method=2, subme=6)),
    (176, 144, "IPPP", 7, dict(rd=0, trellis=0, subme=4, psy=0, psy_rd_q8=0)),
])
def test_weightp_2_blind_duplicate_of_reference_0(w, h, types, seed, over):
    """x264 --weightp 2 on content without fades: every P picture with two or more references carries a duplicate of reference 0 at index 1 with
    the explicit luma weight {1, denom 0, offset -1} (pred_weight_table + list modification naming the picture twice); the decoder's explicit
    weighted prediction (8.4.2.3.2) must land on the encoder's reconstruction, and the duplicate must actually get used"""
    pics = []
    _, order = run(w, h, types, seed, weightp=2, pics_out=pics, **over)
    dupes = used = 0
    for pic, mbs in pics:
        if pic.blind_dupe > 0:
            dupes += 1
            assert pic.slot[0][0] == pic.slot[0][1] and pic.wl0[1].on and pic.wl0[1].offset == -1 and pic.wl0[1].scale == 1 and pic.wl0[1].denom == 0
            inter = (mbs["type"] == O.MB_P_L0) | (mbs["type"] == O.MB_P_8x8)
            used += int(((mbs["ref"] == 1) & inter[..., None]).sum())
    assert dupes >= 2 and used > 0


def test_without_pyramid_every_b_is_disposable():
    _, order = run(176, 144, "IBBBPBBP", 9, pyramid=0, dpb=3)
    assert all(t != 3 for _, t in order)


def test_schedule_is_x264s_coding_order():
    assert bgop.schedule("IBBBP") == [(0, 0), (4, 2), (2, 3), (1, 4), (3, 4)]
    assert bgop.schedule("IBBPBP") == [(0, 0), (3, 2), (1, 3), (2, 4), (5, 2), (4, 4)]


def _host_b_session(tmp_path, n, opts, w=176, h=144, seed=3, inflight=None):
    """x264_encoder_encode() of the product's host library over the stand-in device (tests/stub: the oracle behind the B3 ABI), in a child process"""
    import json
    import os
    import subprocess
    import sys
    out = str(tmp_path / "b.h264")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.pop("X264GPU_INFLIGHT", None)
    if inflight is not None: env["X264GPU_INFLIGHT"] = str(inflight)          # pictures of the session in flight (0: one picture a call, x264's delay without frame threads)
    r = subprocess.run([sys.executable, os.path.join(here, "stub", "run_host_b.py"), out, str(w), str(h), str(n), str(seed)] + opts, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), open(out, "rb").read()


def test_host_session_with_b_pictures(tmp_path):
    """preset medium at constant quantiser through x264_encoder_encode: B / BREF pictures leave in coding order with x264's types and a monotone
    dts (consumers: output/matroska.c:199-202, codec.c:1822-1826), and the stream decodes to every source picture"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 14, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["qp=23", "keyint=30", "scenecut=0", "b-adapt=0"])
    assert (info["bframes"], info["pyramid"], info["badapt"], info["weightb"]) == (3, 2, 0, 1)
    assert info["weightp"] == 2                                                                     # medium's --weightp 2 runs (blind duplicate)
    recs = info["recs"]
    assert len(recs) == n
    TYPE = {1: "I", 3: "P", 4: "R", 5: "B"}
    assert "".join(TYPE[r[0]] for r in recs) == "IPRBBPRBBPRBBP"
    assert [r[1] for r in recs] == [0, 4, 2, 1, 3, 8, 6, 5, 7, 12, 10, 9, 11, 13]
    dts = [r[2] for r in recs]
    assert dts == sorted(dts) and len(set(dts)) == n and all(d <= p for _, p, d, _, _ in recs), "dts must be monotone and never after pts"
    assert dts[:3] == [-2, -1, 0]
    dec = O.h264_decode(stream, n, w, h)
    pocs = O.h264_last_pocs()
    assert [p // 2 for p in pocs] == [r[1] for r in recs]
    frames = synth_frames(w, h, n, seed=3)
    from synth import psnr
    for d, p in zip(dec, pocs):
        assert psnr(d[:w * h], frames[p // 2][:w * h]) > 34.0


@pytest.mark.parametrize("opts,expect", [
    (["preset=slower", "qp=24", "keyint=20", "ref=4"], dict(subme=9, cabac=1, badapt=2, direct=3)),          # x264's slower: subme 9, b-adapt 2, direct auto, umh, trellis 2
    (["no-cabac", "qp=24", "keyint=20", "b-adapt=1", "rc-lookahead=8"], dict(subme=7, cabac=0, badapt=1, direct=1)),      # medium --no-cabac keeps its B pictures (RD on CAVLC counts)
])
def test_host_session_levels_opened_this_round(tmp_path, opts, expect):
    """sessions at the levels round 4 opened — preset slower as asked (--subme 9: RD refinement in B slices, deblock-aware RD) and medium without CABAC
    with its B pictures — through x264_encoder_encode over the stand-in device: the effective parameters say so, B pictures are coded, and the
    stream decodes to every source picture"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 14, 128, 96
    info, stream = _host_b_session(tmp_path, n, opts, w, h, seed=6)
    for k, v in expect.items():
        assert info[k] == v, (k, info[k], v)
    assert info["bframes"] == 3 and any(r[0] in (4, 5) for r in info["recs"]), "B pictures expected"
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n
    frames = synth_frames(w, h, n, seed=6)
    from synth import psnr
    order = [r[1] for r in info["recs"]]
    assert sorted(order) == list(range(n))
    assert min(psnr(dec[k][:w * h], frames[order[k]][:w * h]) for k in range(n)) > 30.0


def test_host_session_weightp_2_without_b_pictures(tmp_path):
    """--bframes 0 --weightp 2: the session runs on the DPB model with no delay (dts = pts, POC type 2), P pictures carry pred_weight_table and
    the duplicate; --weightp 1 (fade analysis only) and sessions that need the other path (mbtree) report weightp 0"""
    n, w, h = 9, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["qp=23", "keyint=30", "scenecut=0", "bframes=0"])
    assert (info["bframes"], info["weightp"]) == (0, 2)
    recs = info["recs"]
    assert [r[0] for r in recs] == [1] + [3] * (n - 1) and [r[1] for r in recs] == list(range(n)) and [r[2] for r in recs] == list(range(n))
    dec = O.h264_decode(stream, n, w, h)
    frames = synth_frames(w, h, n, seed=3)
    from synth import psnr
    for d, f in zip(dec, frames):
        assert psnr(d[:w * h], f[:w * h]) > 34.0
    info0, stream0 = _host_b_session(tmp_path, n, ["qp=23", "keyint=30", "scenecut=0", "bframes=0", "weightp=0"])
    assert info0["weightp"] == 0 and stream0 != stream
    info1, _ = _host_b_session(tmp_path, 3, ["qp=23", "bframes=0", "weightp=1"])
    assert info1["weightp"] == 0
    info2, _ = _host_b_session(tmp_path, 3, ["crf=23", "bframes=0", "rc-lookahead=10"])
    assert (info2["weightp"], info2["mbtree"]) == (0, 1)


def test_host_session_crf_with_b_pictures_and_keyframes(tmp_path):
    """CRF + a short keyint: runs cut short in front of keyframes, the picture before an IDR is never B, every picture decodes"""
    n, w, h = 23, 128, 96
    info, stream = _host_b_session(tmp_path, n, ["crf=24", "keyint=9", "min-keyint=9", "bframes=2", "scenecut=0", "b-adapt=0"], w, h, seed=8)
    recs = info["recs"]
    assert len(recs) == n and info["bframes"] == 2
    by_pts = sorted(recs, key=lambda r: r[1])
    for a, b in zip(by_pts, by_pts[1:]):
        assert not (b[0] == 1 and a[0] in (4, 5)), "a B picture in front of an IDR picture"
    assert [r[0] for r in by_pts][::9] == [1, 1, 1]
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n


def _types_by_display(recs):
    TYPE = {1: "I", 2: "i", 3: "P", 4: "R", 5: "B"}
    return "".join(TYPE[r[0]] for r in sorted(recs, key=lambda r: r[1]))


def test_host_session_b_adapt_1_and_scenecut(tmp_path):
    """medium's own lookahead settings (--b-adapt 1, --scenecut 40) through x264_encoder_encode: the slice types come from
    x264_slicetype_analyse restated on the frame costs of (p0, p1, b) triples (oracle/slicetype.c behind the stub) — a scene cut becomes a
    keyframe at the cut (I inside min-keyint, else IDR), the picture in front of it is never B, runs of B pictures are at most --bframes long,
    and the stream decodes to every source picture"""
    n, w, h = 20, 176, 144
    frames = synth_frames(w, h, n, seed=5, scene_len=11)              # a scene cut at display pictures 11
    info, stream = _host_b_session(tmp_path, n, ["qp=23", "keyint=60", "min-keyint=4", "scene_len=11"], w, h, seed=5)
    assert (info["bframes"], info["badapt"], info["weightp"]) == (3, 1, 2)
    recs = info["recs"]
    assert len(recs) == n
    t = _types_by_display(recs)
    assert t[0] == "I" and t[11] == "I", t                               # the IDR at the start and the one at the cut (11 >= min-keyint)
    assert t[10] in "PI" and t[-1] in "PI", t                            # never a B picture in front of a keyframe or last
    assert "BBBB" not in t.replace("R", "B"), t
    dec = O.h264_decode(stream, n, w, h)
    pocs = O.h264_last_pocs()
    from synth import psnr
    by_coding = [r[1] for r in recs]
    for d, disp in zip(dec, by_coding):
        assert psnr(d[:w * h], frames[disp][:w * h]) > 33.0
    assert len(pocs) == n


def test_host_session_b_adapt_1_static_content_uses_b_pictures(tmp_path):
    """on a nearly static clip b-adapt 1 keeps runs of B pictures, never longer than --bframes.  (What it does picture by picture — "..BP" against "..PP" by path cost —
    is compared with the decision twin in tests/test_decisions_cpu.py; a clip that changes completely from picture to picture is scenecut's business, not this decision's:
    B costs do not consider intra blocks and are scaled by 100 / 120, so with --scenecut 0 such a clip gets B pictures too.)"""
    n, w, h = 13, 128, 96
    info, _ = _host_b_session(tmp_path, n, ["qp=26", "keyint=60", "scenecut=0", "static=1"], w, h, seed=2)
    t = _types_by_display(info["recs"])
    assert t.replace("R", "B").count("B") >= 6 and "BBBB" not in t.replace("R", "B"), t


@pytest.mark.parametrize("seed,extra", [(2, []), (7, ["scene_len=9"]), (4, ["static=1"])])
def test_host_session_b_adapt_2_is_the_cheapest_path(tmp_path, seed, extra):
    """--b-adapt 2 (presets slower and up): x264 slicetype_path, a Viterbi search over the lengths of the window, must find the cheapest way to
    code the window in runs of at most --bframes B pictures — checked against an exhaustive enumeration of every such path on the same frame
    costs (oracle/slicetype.c) for the first decision of the session; the stream decodes"""
    n, w, h = 16, 128, 96
    info, stream = _host_b_session(tmp_path, n, ["qp=26", "keyint=250", "scenecut=0", "b-adapt=2", "no-weightb", "weightp=0"] + extra, w, h, seed=seed, inflight=0)
    assert (info["bframes"], info["badapt"]) == (3, 2)
    assert info["first_output_after"] == 13                              # x264 h->frames.i_delay = max(bframes, 3) * 4 pictures ahead
    # ... + up to i_thread_frames - 1 with pictures in flight (x264 encoder.c: i_delay counts the frame threads; here: until four pictures are out); the stream is the same
    info4, stream4 = _host_b_session(tmp_path, n, ["qp=26", "keyint=250", "scenecut=0", "b-adapt=2", "no-weightb", "weightp=0"] + extra, w, h, seed=seed)
    assert 13 < info4["first_output_after"] <= 13 + 3 and stream4 == stream and info4["recs"] == info["recs"]
    t = _types_by_display(info["recs"]).replace("R", "B")
    assert "BBBB" not in t and t[-1] == "P"
    assert len(O.h264_decode(stream, n, w, h)) == n
    # the window of the first decision: the I picture and the 13 pictures behind it
    frames = synth_frames(w, h, n, seed=seed, **({"scene_len": 9} if extra == ["scene_len=9"] else {}))
    if extra == ["static=1"]:
        rng = np.random.default_rng(seed)
        frames = [np.clip(frames[0].astype(np.int16) + rng.integers(-1, 2, frames[0].shape), 0, 255).astype(np.uint8) for _ in range(n)]
    st = O.OracleSlicetype(w, h, slots=16, bframes=3, weightb=0)
    for i in range(14):
        st.put(i, frames[i])
    memo = {}

    def c(p0, p1, b):
        if (p0, p1, b) not in memo:
            memo[(p0, p1, b)] = st.cost(p0, p1, b, b - p0, p1 - b)
        return memo[(p0, p1, b)]

    def run_cost(a, b_):                                                 # pictures a+1 .. b_ coded as B ... B P behind the non-B picture a
        cost = c(a, b_, b_)
        if b_ - a > 2:
            m = a + (b_ - a) // 2
            cost += c(a, b_, m) + sum(c(a, m, i) for i in range(a + 1, m)) + sum(c(m, b_, i) for i in range(m + 1, b_))
        else:
            cost += sum(c(a, b_, i) for i in range(a + 1, b_))
        return cost

    def compositions(total):
        if total == 0:
            yield ()
        for r in (1, 2, 3, 4):
            if r <= total:
                for rest in compositions(total - r):
                    yield (r,) + rest

    best = None
    for runs in compositions(13):
        pos, cost = 0, 0
        for r in runs:
            cost += run_cost(pos, pos + r)
            pos += r
        if best is None or cost < best[0]:
            best = (cost, runs)
    first_run = t[1:].index("P") + 1
    assert first_run == best[1][0], (t, best)


def test_host_session_crf_on_the_lookaheads_costs(tmp_path):
    """CRF with medium's lookahead (b-adapt 1, scenecut): the quantisers follow from the frame costs of the decided types (x264_rc_analyse_slice:
    the I cost, or the P cost against the last non-B picture); quantisers move with the content and every picture decodes"""
    n, w, h = 16, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["crf=24", "keyint=60", "no-mbtree"], w, h, seed=6)
    assert (info["badapt"], info["mbtree"]) == (1, 0)
    dec = O.h264_decode(stream, n, w, h)
    frames = synth_frames(w, h, n, seed=6)
    from synth import psnr
    for d, r in zip(dec, info["recs"]):
        assert psnr(d[:w * h], frames[r[1]][:w * h]) > 30.0


def test_host_session_mbtree_through_b_pictures(tmp_path):
    """the driver's default rate control — CRF with macroblock-tree and AQ over rc-lookahead pictures — in a session with B pictures: the first
    picture leaves after rc-lookahead + 1 pictures have arrived, the tree (x264 macroblock_tree over the decided types, through B pictures and
    B-references) moves the quantisers of the P / I / B-reference pictures, and the stream decodes"""
    n, w, h = 26, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["crf=24", "keyint=60", "rc-lookahead=10"], w, h, seed=4, inflight=0)
    assert (info["mbtree"], info["badapt"], info["bframes"]) == (1, 1, 3)
    assert info["first_output_after"] == 11
    info4, stream4 = _host_b_session(tmp_path, n, ["crf=24", "keyint=60", "rc-lookahead=10"], w, h, seed=4)          # four pictures in flight: three more calls of delay
    assert 11 < info4["first_output_after"] <= 11 + 3 and stream4 == stream and info4["recs"] == info["recs"]
    dec = O.h264_decode(stream, n, w, h)
    frames = synth_frames(w, h, n, seed=4)
    from synth import psnr
    for d, r in zip(dec, info["recs"]):
        assert psnr(d[:w * h], frames[r[1]][:w * h]) > 29.0
    info0, stream0 = _host_b_session(tmp_path, n, ["crf=24", "keyint=60", "rc-lookahead=10", "no-mbtree"], w, h, seed=4)
    assert info0["mbtree"] == 0 and stream0 != stream


def _batch(n, w, h, nf, opts, gpu=False, timeout=900, overlap=True):
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    env.pop("X264GPU_BATCH", None)
    env["X264GPU_BATCH_OVERLAP"] = "1" if overlap else "0"          # the host's entropy coding beside the device's next round (one picture of delay), or inside the call
    r = subprocess.run([sys.executable, os.path.join(here, "stub", "run_host_batch.py"), str(n), str(w), str(h), str(nf)] + opts + (["--gpu"] if gpu else []),
                       capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("n,opts", [
    (4, ["qp=23", "keyint=30", "scenecut=0", "b-adapt=0", "bframes=3"]),                     # medium's B structure, constant quantisers
    (3, ["crf=24", "keyint=8", "min-keyint=8", "scenecut=0", "b-adapt=0", "bframes=2", "no-mbtree"]),      # CRF: every stream its own quantisers
    (2, ["qp=26", "keyint=30", "scenecut=0", "bframes=0", "weightp=2"]),                      # no B pictures
])
def test_cross_session_batcher_is_byte_identical(n, opts):
    """X264GPU_BATCH=N: N x264_encoder_open sessions driven from N host threads share one device encoder (N streams, one lock-step launch per
    picture); every session's stream equals the one it writes on its own (driverproc.c:110-128: one CODEC per stream)"""
    r = _batch(n, 176, 144, 11, opts)
    assert r["equal"] == [True] * n and r["distinct"] == n, r
    r0 = _batch(n, 176, 144, 11, opts, overlap=False)          # ... with download and entropy coding inside the call: the same streams
    assert r0["equal"] == [True] * n and r0["sizes"] == r["sizes"], (r0, r)


def fade_frames(w, h, n, seed, step=6):
    fr = synth_frames(w, h, n, seed=seed)
    out = []
    for i, f in enumerate(fr):
        a = max(0.0, 1.0 - i * step / 100.0)
        g = f.astype(np.float32)
        g[:w * h] *= a
        g[w * h:] = 128 + (g[w * h:] - 128) * a
        out.append(np.clip(np.rint(g), 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("types,weights,weightp,over", [
    ("IPPPP", {1: (60, 6, 0), 2: (59, 6, 1), 3: (15, 4, -2), 4: (1, 0, -3)}, 2, {}),              # weighted reference 0 + both duplicates
    ("IBBPBBP", {3: (53, 6, 1), 6: (111, 7, 0)}, 2, {}),
    ("IPPP", {1: (60, 6, 0), 2: (59, 6, 1), 3: (29, 5, 2)}, 1, {}),                               # --weightp 1: the weight alone
    ("IPPP", {1: (1, 0, -128), 2: (60, 6, 0)}, 2, dict(refs=1)),                                  # offset -128 has no duplicate one step down; one reference: no duplicates at all
    # ... with the chroma planes weighted beside luma (x264_weights_analyse once luma has a weight): (.., chroma denom, Cb on / scale / offset, Cr ...)
    ("IPPP", {1: (58, 6, 3, 5, 1, 30, 4, 1, 29, 6)}, 2, {}),
    ("IBPBP", {2: (60, 6, -2, 6, 1, 62, -3, 0, 1, 0), 4: (66, 6, 2, 5, 0, 1, 0, 1, 31, 2)}, 2, {}),   # one chroma plane alone: the other is sent as 1 << denom
    ("IPP", {1: (60, 6, 1, 4, 1, 15, 0, 1, 17, -2), 2: (1, 0, 4, 6, 1, 60, 1, 1, 61, 0)}, 1, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),
    ("IPPP", {2: (62, 6, -1, 6, 1, 60, 2, 1, 66, -4)}, 2, dict(subme=8, rd=63, me_method=2)),          # RD refinement: the part costs' chroma predictions carry the weights
])
def test_explicit_luma_weights_of_reference_0(types, weights, weightp, over):
    """the weights x264_weights_analyse gives a P picture of a fade: reference 0 carries {scale, denom, offset}; under --weightp 2 its duplicate one
    offset step down and an unweighted duplicate follow (x264 weighted_reference_duplicate twice); pred_weight_table with a non-zero denominator,
    explicit weighted prediction in the checker decoder, the loop filter seeing one picture behind three indices"""
    w, h = 176, 144
    pics = []
    run(w, h, types, 3, weightp=weightp, pics_out=pics, weights=weights, frames=fade_frames(w, h, len(types), 3), **over)
    seen = 0
    for pic, mbs in pics:
        if pic.wl0[0].on:
            seen += 1
            n0 = pic.nref[0]
            if weightp == 2 and n0 >= 3:
                assert pic.slot[0][0] == pic.slot[0][1] == pic.slot[0][2] or pic.wl0[0].offset == -128
    assert seen == len(weights)


@pytest.mark.parametrize("opts,per_pic", [(["slices=4"], 4), (["sliced-threads", "threads=2"], 2)])
def test_host_session_b_pictures_in_slices(tmp_path, opts, per_pic):
    """--slices N / x264's slice threads in a session with B pictures (the driver's default rate control: CRF + AQ + macroblock-tree): every picture,
    B pictures too, leaves as `per_pic` slice NAL units; slice threads switch the loop filter off across slices (disable_deblocking_filter_idc 2);
    the stream decodes to the source"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 16, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["crf=24", "keyint=12"] + opts)
    assert (info["bframes"], info["mbtree"], info["weightp"]) == (3, 1, 2)
    recs = info["recs"]
    assert len(recs) == n and any(r[0] in (4, 5) for r in recs)
    _, _, sl = O.lsmash_parse(stream, max_slices=256)
    assert len(sl) == n * per_pic
    k = 0
    for r in recs:
        want = {1: 2, 2: 2, 3: 0, 4: 1, 5: 1}[r[0]]
        assert all(x.slice_type % 5 == want for x in sl[k:k + per_pic]), (r, [x.slice_type for x in sl[k:k + per_pic]])
        k += per_pic
    dec = O.h264_decode(stream, n, w, h)
    frames = synth_frames(w, h, n, seed=3)
    from synth import psnr
    for d, r in zip(dec, recs):
        assert psnr(d[:w * h], frames[r[1]][:w * h]) > 30.0


def test_host_session_b8x8_reaches_the_b_slices(tmp_path):
    """--partitions b8x8 (medium has it) must reach the analysis of B slices through x264_encoder_open: the stream with it differs from the stream
    without it, from the first B picture on"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 10, 176, 144
    base = ["qp=24", "keyint=30", "scenecut=0", "b-adapt=0"]
    a, sa = _host_b_session(tmp_path, n, base + ["partitions=p8x8,b8x8,i8x8,i4x4"], w, h, seed=9)
    b, sb = _host_b_session(tmp_path, n, base + ["partitions=p8x8,i8x8,i4x4"], w, h, seed=9)
    assert sa != sb
    assert [r[:4] for r in a["recs"]] == [r[:4] for r in b["recs"]]
    assert [r[4] for r in a["recs"][:2]] == [r[4] for r in b["recs"][:2]]          # the I and the first P picture are coded before any B picture
    assert any(ra[4] != rb[4] for ra, rb in zip(a["recs"], b["recs"]) if ra[0] in (4, 5))


def test_host_session_says_what_it_does_not_run(tmp_path):
    """options for tools this path has not got are accepted (the driver passes the user's command line through x264_param_parse, codec.c:1349) and
    reported through pf_log (the driver's log window, codec.c:1274-1283) with what runs instead — never dropped silently"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    info, _ = _host_b_session(tmp_path, 3, ["log=1", "crf=24", "vbv-maxrate=2000", "vbv-bufsize=2000", "nr=200", "slice-max-size=1500", "fake-interlaced", "bluray-compat",
                                            "direct=none", "subme=9", "psy-rd=1.0:0.2", "open-gop", "ref=9", "me=tesa"], 64, 48)
    text = " | ".join(m for lvl, m in info["log"] if lvl <= 2)
    for needle in ("VBV", "nr (noise reduction)", "slice-max-size", "fake-interlaced", "bluray-compat", "direct", "subme", "psy-trellis", "open-gop", "ref %d -> 5", "tesa"):
        assert needle in text, (needle, text)


def test_host_session_b_bias_moves_the_b_decisions(tmp_path):
    """--b-bias: x264's i_bframe_bias enters b-adapt 1's thresholds AND the scaling of every B cost in slicetype_frame_cost (100 / (120 + bias)):
    a positive bias never gives fewer B pictures than a negative one"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    counts = {}
    for bias in (-50, 0, 60):
        info, _ = _host_b_session(tmp_path, 14, ["qp=26", "keyint=60", "scenecut=0", f"b-bias={bias}"], 128, 96, seed=2)
        counts[bias] = _types_by_display(info["recs"]).replace("R", "B").count("B")
    assert counts[-50] <= counts[0] <= counts[60] and counts[-50] < counts[60], counts


def test_host_session_single_pass_abr_with_b_pictures(tmp_path):
    """--bitrate N (x264vfw's single-pass ABR page, config.c / codec.c:x264vfw 'Single pass - bitrate-based (ABR)') keeps B pictures, the
    lookahead and macroblock-tree: the coded size of every picture, B pictures' divided by pbratio, steers the rate factor
    (x264_ratecontrol_end), the stream lands near the target and decodes to the source"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 75, 176, 144
    rates = {}
    for kbps in (200, 600):
        info, stream = _host_b_session(tmp_path, n, [f"bitrate={kbps}", "keyint=50"])
        assert (info["bframes"], info["mbtree"], info["weightp"]) == (3, 1, 2)
        recs = info["recs"]
        assert len(recs) == n and any(r[0] in (4, 5) for r in recs)
        rates[kbps] = sum(r[4] for r in recs) * 8 / n * 25 / 1000
        assert abs(rates[kbps] - kbps) < 0.2 * kbps, rates
        dec = O.h264_decode(stream, n, w, h)
        assert len(dec) == n
        frames = synth_frames(w, h, n, seed=3)
        from synth import psnr
        for d, r in zip(dec, recs):                                       # decode order = coding order; pts = source picture
            assert psnr(d[:w * h], frames[r[1]][:w * h]) > (30.0 if kbps == 600 else 24.0)
    assert rates[600] > 2 * rates[200]


def test_host_session_finds_the_weights_of_a_fade(tmp_path):
    """a clip that fades to black through x264_encoder_encode with medium's lookahead: x264_weights_analyse restated on the lookahead's primitives
    gives the P pictures luma weights (the stream shrinks against --weightp 0 and decodes to the source)"""
    n, w, h = 14, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["qp=23", "keyint=60", "fade=6"], w, h, seed=3)
    assert info["weightp"] == 2
    info0, stream0 = _host_b_session(tmp_path, n, ["qp=23", "keyint=60", "fade=6", "weightp=0"], w, h, seed=3)
    assert len(stream) < 0.97 * len(stream0), (len(stream), len(stream0))
    dec = O.h264_decode(stream, n, w, h)
    wl, wc = O.h264_last_weighted()
    assert wl >= 3 and wc >= 2, (wl, wc)          # the fade scales the chroma planes towards grey: x264 weights them beside luma
    frames = fade_frames(w, h, n, 3)
    from synth import psnr
    for d, r in zip(dec, info["recs"]):
        assert psnr(d[:w * h], frames[r[1]][:w * h]) > 33.0
        assert psnr(d[w * h:], frames[r[1]][w * h:]) > 33.0


def test_host_session_two_pass(tmp_path):
    """x264's 2-pass rate control (the driver's encoding type 4, codec.c:1516-1541): pass 1 (ABR) writes one statistics line per coded picture — type,
    quantiser, bits split into texture / vectors / the rest — and pass 2 plans every picture's quantiser from them (init_pass2 restated): it codes the
    same picture types and lands within 5 % of the requested size; the stream decodes"""
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h, kbps = 75, 176, 144, 260
    st = str(tmp_path / "x264.stats")
    common = ["scene_len=31", f"bitrate={kbps}", "keyint=40", "bframes=3", "subme=5", "trellis=0", "log=1", f"stats={st}"]
    info1, _ = _host_b_session(tmp_path, n, common + ["pass=1"], w, h, seed=9)
    lines = [ln for ln in open(st).read().splitlines() if not ln.startswith("#")]
    assert len(lines) == n and all(ln.startswith("in:") and " tex:" in ln and " mv:" in ln and " misc:" in ln for ln in lines)
    assert not os.path.exists(st + ".temp")
    info2, stream = _host_b_session(tmp_path, n, common + ["pass=2"], w, h, seed=9)
    assert any("planned from the first pass" in m for _, m in info2["log"])
    assert [(r[0], r[1]) for r in info2["recs"]] == [(r[0], r[1]) for r in info1["recs"]]          # the first pass' picture types, in its coding order
    rate = len(stream) * 8 / (n / 25.0) / 1000.0
    assert abs(rate / kbps - 1.0) < 0.05, rate
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n
    assert [ln for ln in open(st).read().splitlines() if not ln.startswith("#")] == lines          # pass 2 leaves the statistics alone ...
    # ... the N-th pass (--pass 3: the driver's encoding type 4 past the first pass, codec.c:1519-1541 with its updatestats) reads them AND writes this pass' own
    # lines in their place: same picture types, the quantisers and sizes of THIS pass, and a further pass plans from those
    info3, stream3 = _host_b_session(tmp_path, n, common + ["pass=3"], w, h, seed=9)
    assert stream3 == stream and not os.path.exists(st + ".temp")
    lines3 = [ln for ln in open(st).read().splitlines() if not ln.startswith("#")]
    field = lambda ln, k: ln.split(" " + k + ":")[1].split(" ")[0]
    assert len(lines3) == n and [field(" " + ln, "type") for ln in lines3] == [field(" " + ln, "type") for ln in lines]
    assert lines3 != lines and abs(sum(int(field(ln, "tex")) + int(field(ln, "mv")) + int(field(ln, "misc")) for ln in lines3) - len(stream) * 8) <= 8 * n
    info4, stream4 = _host_b_session(tmp_path, n, common + ["pass=2"], w, h, seed=9)
    assert abs(len(stream4) * 8 / (n / 25.0) / 1000.0 / kbps - 1.0) < 0.05
    # a bitrate the headers alone exceed is refused the way x264 refuses it
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub", "run_host_b.py"), str(tmp_path / "c.h264"), str(w), str(h), str(n), "9"] +
                       ["scene_len=31", "bitrate=1", "keyint=40", "bframes=3", "pass=2", f"stats={st}"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


@pytest.mark.parametrize("mode", [2, 3])
def test_host_session_aq_mode_2_and_3(tmp_path, mode):
    """--aq-mode 2 / 3 (config.c:1610-1614 passes the user's choice through): the session computes x264_adaptive_quant_frame's auto-variance offsets
    when a picture arrives and hands them to the encoder; the macroblock quantisers differ from mode 1's and the stream decodes"""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", __file__.rsplit("/", 1)[0] + "/stub"])
    n, w, h = 10, 176, 144
    info, stream = _host_b_session(tmp_path, n, ["crf=24", "keyint=30", f"aq-mode={mode}", "log=1", "no-mbtree"], w, h, seed=6)
    assert not any("aq-mode" in m and "aq-mode 1" in m for _, m in info["log"])
    info1, stream1 = _host_b_session(tmp_path, n, ["crf=24", "keyint=30", "aq-mode=1", "no-mbtree"], w, h, seed=6)
    assert stream != stream1
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n

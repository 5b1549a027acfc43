"""-m gpu: the drop-in boundary B1 end to end — x264_param_default_preset -> x264_encoder_open ->
x264_encoder_encode loop (the exact call sequence of codec.c:1463,1623,1693,1848-1857) on the MI355X path,
bitstream checked by the decoder and against the bitstream the oracle's records produce."""
import ctypes as C
import os

import numpy as np
import pytest

import host_lib as HL
import oracle_lib as O
from synth import psnr, synth_frames

pytestmark = pytest.mark.gpu
H = HL.H


def open_encoder(w, h, opts, profile=b"baseline", preset=b"medium"):
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), preset, None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den = 25, 1
    p.i_log_level = -1
    if "bframes" not in opts:
        opts = dict(opts, bframes=0)          # these sessions test the I / P behaviours (one picture back per call); B sessions: test_b_session_*
        opts.setdefault("weightp", 0)         # ... of the ring path (--weightp 2 moves a session without B pictures onto the DPB model: test_weightp_session_*)
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, k
    p.b_vfr_input = 0
    p.b_annexb, p.b_repeat_headers = 1, 1                              # VfW mode (codec.c:1611-1615)
    assert H.x264_param_apply_profile(C.byref(p), profile) == 0        # profile None = no restriction
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_, "x264_encoder_open failed"
    eff = HL.Param()
    H.x264_encoder_parameters(h_, C.byref(eff))
    return h_, eff


def eff_kw(eff):
    """the oracle's config fields that follow from the session's effective parameters (x264_encoder_parameters): sub-pel level, RD mode
    decision + psy (subme 6 / 7 in --no-cabac sessions; the chroma quantiser offset the encoder reports already carries x264's psy
    compensation), entropy coder, vector range"""
    rd = int(eff.analyse.i_subpel_refine >= 6)
    q8 = int(eff.analyse.f_psy_rd * 256.0 + 0.5) if rd and eff.analyse.b_psy else 0
    return dict(subme=eff.analyse.i_subpel_refine, rd=rd, psy=int(rd and eff.analyse.b_psy), psy_rd_q8=q8, trellis=(63 + 64 if eff.analyse.i_trellis == 2 else 63) if eff.analyse.i_trellis else 0, chroma_qp_offset=eff.analyse.i_chroma_qp_offset,
                mv_range=eff.analyse.i_mv_range, cabac=eff.b_cabac)


def encode_all(h_, w, h, frames):
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    stream, info, recons = b"", [], []
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)              # planes are contiguous in x264_picture_alloc
        pic.i_pts = i
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size > 0
        # all NALs of a call are contiguous from nal[0].p_payload (relied on at codec.c:1703,1719)
        assert sum(nal[k].i_payload for k in range(n.value)) == size
        for k in range(1, n.value):
            assert C.addressof(nal[k].p_payload.contents) == C.addressof(nal[k - 1].p_payload.contents) + nal[k - 1].i_payload
        stream += C.string_at(nal[0].p_payload, size)
        info.append((out.i_type, out.b_keyframe, out.i_pts, [nal[k].i_type for k in range(n.value)]))
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        assert H.x264host_get_recon(h_, rec.ctypes.data) == 0
        recons.append(rec)
    assert H.x264_encoder_delayed_frames(h_) == 0
    assert H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out)) == 0   # flush: nothing delayed
    H.x264_picture_clean(C.byref(pic))
    return stream, info, recons


@pytest.mark.parametrize("w,h,opts", [(176, 144, {"qp": 26, "keyint": 4, "no-scenecut": None}), (352, 288, {"qp": 30, "keyint": 250}),
                                       (208, 120, {"qp": 22, "keyint": 3, "no-deblock": None, "no-scenecut": None}),
                                       (176, 144, {"qp": 28, "keyint": 250, "ref": 1, "partitions": "i4x4"}),
                                       (176, 144, {"qp": 24, "keyint": 250, "_profile": b"high"}),      # 8x8dct: High profile stream
                                       (352, 288, {"qp": 27, "keyint": 5, "no-scenecut": None, "_profile": b"high"})])
def test_encode_api_closed_loop(gpu, w, h, opts):
    nfr = 7
    frames = synth_frames(w, h, nfr, seed=w + 3 * h)
    opts = dict(opts)
    profile = opts.pop("_profile", b"baseline")
    h_, eff = open_encoder(w, h, opts, profile)
    assert eff.analyse.b_transform_8x8 == int(profile == b"high")
    assert (eff.i_bframe, eff.b_cabac, eff.rc.i_rc_method) == (0, int(profile == b"high"), HL.X264_RC_CQP)   # effective params: CABAC stays on above Baseline
    assert eff.i_frame_reference == int(opts.get("ref", 3))                         # medium: --ref 3
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    keyint = opts["keyint"]
    for i, (typ, key, pts, nal_types) in enumerate(info):
        idr = i % keyint == 0
        assert key == int(idr) and pts == i and typ == (1 if idr else 3)
        assert nal_types[-1] == (5 if idr else 1)
        if idr:
            assert nal_types[:2] == [7, 8]                                # SPS/PPS before each keyframe (codec.c:1614)
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i} != encoder reconstruction")


@pytest.mark.parametrize("opts,subme,cqo", [({}, 7, -2), ({"no-psy": None}, 7, 0), ({"psy-rd": "0.1:0", "subme": 6}, 6, -1), ({"subme": 9, "chroma-qp-offset": 3}, 7, 1),
                                            ({"psy-rd": "0:0"}, 7, 0)])       # psy on at strength 0: no energy term, no offset change, but the chroma lambda table applies
def test_rd_session_equals_oracle_pipeline(gpu, opts, subme, cqo):
    """preset medium without CABAC (Baseline profile): subme 7 = RD mode decision on the device with CAVLC bit counts and psy-RD; x264 lowers
    the chroma quantiser offset to compensate psy (by 2, or 1 below psy-rd 0.25), and levels above 7 (RD refinement) come back as 7.  The
    stream decodes to the encoder's reconstruction and the oracle pipeline (RD on) reconstructs the same samples"""
    w, h, nfr, qp = 176, 144, 5, 27
    frames = synth_frames(w, h, nfr, seed=606)
    h_, eff = open_encoder(w, h, dict({"qp": qp, "keyint": 250}, **opts), b"baseline")
    assert (eff.b_cabac, eff.analyse.i_subpel_refine, eff.analyse.i_chroma_qp_offset, eff.analyse.i_trellis) == (0, subme, cqo, 0)      # no CABAC: no trellis
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    kw = eff_kw(eff)
    assert kw["rd"] == 1 and kw["psy"] == int("no-psy" not in opts)
    qp_i = max(1, int(qp - 6.0 * np.log2(1.4) + 0.5))
    og = O.OracleEncoder(O.default_config(w, h, qp_i=qp_i, qp_p=qp, partitions=3, refs=3, chroma_me=1, mixed_refs=1, **kw))
    dec = O.h264_decode(stream, nfr, w, h)
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if i == 0 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


@pytest.mark.parametrize("opts", [{}, {"no-psy": None, "ref": 1}, {"sliced-threads": None, "threads": 2}, {"trellis": 2, "me": "umh", "ref": 4}, {"trellis": 0}])
def test_medium_session_runs_rd_with_cabac(gpu, opts):
    """preset medium as the reference's driver opens it (High profile, CABAC): subme 7 stays 7 — RD mode decision with CABAC sizes + psy-RD on
    the device.  The stream decodes to the encoder's reconstruction and the oracle pipeline (cabac, rd) reconstructs the same samples"""
    w, h, nfr, qp = 176, 144, 5, 26
    frames = synth_frames(w, h, nfr, seed=707)
    h_, eff = open_encoder(w, h, dict({"qp": qp, "keyint": 250}, **opts), b"high")
    assert (eff.b_cabac, eff.analyse.i_subpel_refine, eff.analyse.i_trellis) == (1, 7, opts.get("trellis", 1))        # medium: trellis 1 on the device too; 2 and 0 honoured
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    kw = eff_kw(eff)
    assert kw["rd"] == 1 and kw["cabac"] == 1 and kw["chroma_qp_offset"] == (0 if "no-psy" in opts else -2)
    qp_i = max(1, int(qp - 6.0 * np.log2(1.4) + 0.5))
    og = O.OracleEncoder(O.default_config(w, h, qp_i=qp_i, qp_p=qp, partitions=7, dct8x8=1, refs=eff.i_frame_reference, chroma_me=1, mixed_refs=int(eff.i_frame_reference > 1),
                                          slices=2 if "sliced-threads" in opts else 1, me_method=2 if opts.get("me") == "umh" else 1, **kw))
    dec = O.h264_decode(stream, nfr, w, h)
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if i == 0 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


def test_preset_ultrafast_is_fully_covered(gpu):
    """preset ultrafast (config.c:1460-1466): every tool x264 uses there exists in this path — me dia, subme 0, ref 1,
    no partitions, no 8x8dct, CAVLC, no deblock, no B-frames — so the effective parameters equal the requested ones and the
    stream equals the one entropy-coded from the oracle's records."""
    w, h, nfr, qp = 352, 288, 5, 26
    frames = synth_frames(w, h, nfr, seed=352)
    h_, eff = open_encoder(w, h, {"qp": qp, "keyint": 250}, None, b"ultrafast")
    assert (eff.analyse.i_me_method, eff.analyse.i_subpel_refine, eff.i_frame_reference, eff.b_cabac, eff.i_bframe) == (0, 0, 1, 0, 0)
    assert (eff.analyse.b_transform_8x8, eff.b_deblocking_filter, eff.analyse.inter & 0x10) == (0, 0, 0)
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    qp_i = max(1, int(qp - 6.0 * np.log2(1.4) + 0.5))
    enc = O.OracleEncoder(O.default_config(w, h, qp_i=qp_i, qp_p=qp, partitions=0x100, refs=1, me_method=0, subme=0, deblock=0, mv_range=eff.analyse.i_mv_range, cabac=eff.b_cabac))
    ref = b""
    for i, f in enumerate(frames):
        mbs, lv = enc.encode(f, 2 if i == 0 else 0)
        ref += HL.write_slice(22, 18, 2 if i == 0 else 0, qp_i if i == 0 else qp, qp, i, 8, int(i == 0), 0, 1, mbs, lv)[0]
        np.testing.assert_array_equal(recons[i], enc.recon())
    import re
    nals = lambda b: [n for n in re.split(b"\x00\x00\x00\x01|\x00\x00\x01", b) if n and (n[0] & 31) in (1, 5)]
    assert nals(stream) == nals(ref)
    dec = O.h264_decode(stream, nfr, w, h)
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i])


def test_bitstream_equals_oracle_path(gpu):
    """same records -> same bytes: the GPU path's slice NALs equal the ones coded from the oracle's records"""
    w, h, nfr, qp = 176, 144, 5, 27
    frames = synth_frames(w, h, nfr, seed=99)
    h_, eff = open_encoder(w, h, {"qp": qp, "keyint": 250}, b"high")
    stream, info, _ = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    qp_i = max(1, int(qp - 6.0 * np.log2(1.4) + 0.5))
    enc = O.OracleEncoder(O.default_config(w, h, qp_i=qp_i, qp_p=qp, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, **eff_kw(eff)))      # medium: p8x8 + i4x4 + i8x8, ref 3, 8x8dct, chroma-me; subme 7: RD with CABAC sizes
    ref = b""
    for i, f in enumerate(frames):
        mbs, lv = enc.encode(f, 2 if i == 0 else 0)
        ref += HL.write_slice(11, 9, 2 if i == 0 else 0, qp_i if i == 0 else qp, qp, i, 8, int(i == 0), 0, 0, mbs, lv,
                              num_ref=max(1, min(3, i)), num_ref_default=3, t8x8=1, cabac=1)[0]
    def slice_nals(b):
        import re
        return [n.rstrip(b"\x00") if False else n for n in re.split(b"\x00\x00\x00\x01|\x00\x00\x01", b) if n and (n[0] & 31) in (1, 5)]
    assert slice_nals(stream) == slice_nals(ref)


def test_headers_call(gpu):
    h_, _ = open_encoder(64, 48, {"qp": 30})
    nal, n = C.POINTER(HL.Nal)(), C.c_int()
    size = H.x264_encoder_headers(h_, C.byref(nal), C.byref(n))
    assert n.value == 3 and [nal[k].i_type for k in range(3)] == [7, 8, 6] and size == sum(nal[k].i_payload for k in range(3))
    for k in range(3):                                                       # 4-byte prefix each (output/raw.c:41-47)
        assert bytes(nal[k].p_payload[0:4]) == b"\x00\x00\x00\x01"
    H.x264_encoder_close(h_)


@pytest.mark.parametrize("w,h,nfr,keyint,threads,rc", [(176, 144, 23, 4, 3, "qp"), (96, 80, 17, 5, 4, "qp"), (176, 144, 12, 4, 3, "qp"), (64, 48, 9, 3, 8, "qp"),
                                                          (64, 48, 2, 6, 2, "qp"), (176, 144, 23, 4, 3, "crf"), (96, 80, 14, 5, 4, "crf-noaq"), (64, 48, 9, 3, 8, "crf")])
def test_gop_parallel_equals_serial(gpu, w, h, nfr, keyint, threads, rc):
    """--threads G codes G closed GOPs in lock-step: the frames come out (G-1)*keyint (+1) calls late, in order, and the stream is
    byte-identical to the serial encode (fixed keyint makes the GOPs independent); flush drains the rest.  Under CRF every GOP slot
    carries the quantiser the serial rate control gives that picture (it follows from the lookahead costs alone), with AQ on top."""
    frames = synth_frames(w, h, nfr, seed=31 * w + nfr)
    if rc != "qp":                                                           # a sequence whose complexity moves, so the quantisers do
        frames = [f if i % 5 else synth_frames(w, h, 1, seed=900 + i)[0] for i, f in enumerate(frames)]
    opts = {"keyint": keyint, "min-keyint": keyint, "no-scenecut": None}      # SURVEY config 5: fixed closed GOPs
    opts.update({"qp": 27} if rc == "qp" else {"crf": 25, "no-mbtree": None})
    if rc == "crf-noaq":
        opts["aq-mode"] = 0
    h1, e1 = open_encoder(w, h, opts, b"high")
    assert e1.rc.i_rc_method == (HL.X264_RC_CQP if rc == "qp" else HL.X264_RC_CRF)
    serial, info1, _ = encode_all(h1, w, h, frames)
    H.x264_encoder_close(h1)
    hg, eff = open_encoder(w, h, dict(opts, threads=threads), b"high")
    assert eff.i_threads == threads and eff.rc.i_rc_method == e1.rc.i_rc_method and eff.rc.i_aq_mode == e1.rc.i_aq_mode
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, n = C.POINTER(HL.Nal)(), C.c_int()
    stream, pts_out, calls_with_output = b"", [], 0
    delay = (threads - 1) * keyint
    for i, f in enumerate(frames):
        for pl, (sz, off) in enumerate([(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]):
            C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
        pic.i_pts = 100 + i
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size >= 0
        assert size == 0 or i >= delay, f"call {i}: a frame came out before the {delay}-frame GOP delay"     # (+1: CAVLC overlaps the next call)
        if size:
            stream += C.string_at(nal[0].p_payload, size); pts_out.append(out.i_pts)
        assert H.x264_encoder_delayed_frames(hg) == i + 1 - len(pts_out), (i, size, len(pts_out), H.x264_encoder_delayed_frames(hg))
    while H.x264_encoder_delayed_frames(hg):
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        stream += C.string_at(nal[0].p_payload, size); pts_out.append(out.i_pts)
    assert H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), None, C.byref(out)) == 0
    H.x264_encoder_close(hg)
    H.x264_picture_clean(C.byref(pic))
    assert pts_out == [100 + i for i in range(nfr)]
    assert stream == serial
    if rc != "qp":                                                           # and the serial session's quantisers did move
        h2, _ = open_encoder(w, h, opts, b"high")
        again, rows, _ = encode_with_decisions(h2, w, h, frames)
        H.x264_encoder_close(h2)
        assert again == serial and len({r[1] for r in rows if not r[0]}) > 1, [r[1] for r in rows]
    assert len(O.h264_decode(stream, nfr, w, h)) == nfr


def encode_with_decisions(h_, w, h, frames):
    """encode + flush (the lookahead may hold pictures back: x264_encoder_delayed_frames / encode(NULL), codec.c:1842-1856), with
    the per-picture decisions (quantiser, scenecut flag, lookahead sums) the host took; rows are in output = input order"""
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    stream, rows, recons = b"", [], []

    def take(size, nal):
        nonlocal stream
        stream += C.string_at(nal[0].p_payload, size)
        qp, sc, costs = C.c_int(), C.c_int(), (C.c_int32 * 4)()
        assert H.x264host_last_decision(h_, C.byref(qp), C.byref(sc), costs) == 0
        assert out.i_pts == len(rows)
        rows.append((int(out.b_keyframe), qp.value, sc.value, list(costs), int(out.i_type), float(H.x264host_last_qpm(h_))))          # [5]: the float quantiser (x264 rc->qpm) handed to the device
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        assert H.x264host_get_recon(h_, rec.ctypes.data) == 0
        recons.append(rec)
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
        pic.i_pts = i
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size >= 0
        if size:
            take(size, nal)
        assert H.x264_encoder_delayed_frames(h_) == i + 1 - len(rows)
    while H.x264_encoder_delayed_frames(h_):
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        take(size, nal)
    assert len(rows) == len(frames)
    H.x264_picture_clean(C.byref(pic))
    return stream, rows, recons


def test_scenecut_inserts_idr(gpu):
    """--scenecut 40 (x264 default): a hard cut becomes an IDR picture (lookahead P cost ~ intra cost), continuous content does
    not; the stream decodes to the encoder's reconstruction and equals the oracle pipeline driven by the same slice types."""
    w, h = 176, 144
    frames = synth_frames(w, h, 6, seed=5) + synth_frames(w, h, 5, seed=99)
    h_, eff = open_encoder(w, h, {"qp": 27, "keyint": 250, "min-keyint": 2}, b"high")
    assert eff.i_scenecut_threshold == 40 and eff.i_keyint_min == 2
    stream, rows, recons = encode_with_decisions(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert [r[0] for r in rows] == [1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0]
    assert rows[6][2] == 1 and rows[6][3][1] >= 0.9 * rows[6][3][0] and all(r[2] == 0 for i, r in enumerate(rows) if i != 6)
    dec = O.h264_decode(stream, len(frames), w, h)
    og = O.OracleEncoder(O.default_config(w, h, qp_i=24, qp_p=27, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, **eff_kw(eff)))
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if rows[i][0] else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


def test_crf_follows_the_lookahead_complexity(gpu):
    """--crf (the driver's default rate control, codec.c:1504-1507) without AQ / mbtree: one quantiser per picture from the
    lookahead cost ([x264-upstream] ratecontrol.c rate_estimate_qscale, CRF branch, restated here).  The decoder reproduces the
    reconstruction (slice_qp_delta carries the quantiser) and the oracle pipeline fed the same quantisers matches too."""
    w, h, crf, qcomp, ipf = 176, 144, 26.0, 0.6, 1.4
    frames = synth_frames(w, h, 5, seed=5) + synth_frames(w, h, 4, seed=99)
    h_, eff = open_encoder(w, h, {"crf": crf, "keyint": 250, "min-keyint": 3, "no-mbtree": None}, b"high")
    assert eff.rc.i_rc_method == HL.X264_RC_CRF and eff.rc.b_mb_tree == 0 and eff.rc.i_lookahead == 0
    stream, rows, recons = encode_with_decisions(h_, w, h, frames)
    H.x264_encoder_close(h_)
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    q2s, s2q = (lambda q: 0.85 * 2.0 ** ((q - 12.0) / 6.0)), (lambda s: 12.0 + 6.0 * np.log2(s / 0.85))
    rfc = (nmb * 80.0) ** (1 - qcomp) / q2s(crf)
    cs = cc = 0.0
    apn = 0.01; apq = crf * apn                                  # x264_ratecontrol_new: the running P quantiser starts with a hundredth of a picture at ABR_INIT_QP
    last_i = True
    qps = []
    for i, (key, qp, sc, costs, _typ, qpm) in enumerate(rows):
        satd = costs[0] if key else costs[1]
        cs, cc = cs * 0.5 + satd, cc * 0.5 + 1
        q = (cs / cc) ** (1 - qcomp) / rfc
        if key and not last_i:
            q = q2s(apq / apn) / ipf
        elif i == 0:
            q = q2s(crf) / ipf                                   # very first picture: ABR_INIT_QP / ipratio
        qpf = float(np.clip(s2q(q), 1, 51))
        apq, apn = apq * 0.95 + (qpf + 6 * np.log2(ipf) if key else qpf), apn * 0.95 + 1
        last_i = bool(key)
        qps.append(int(qpf + 0.5))
    assert [r[1] for r in rows] == qps, (qps, rows)
    assert all(abs(r[5] - r[1]) <= 0.5 + 1e-6 and r[5] != 0 for r in rows)          # ... each the rounding of the float quantiser the macroblock quantisers start from
    assert len(set(qps)) > 1 and rows[5][0] == 1                     # the cut is an IDR and the quantiser moves with the content
    dec = O.h264_decode(stream, len(frames), w, h)
    assert eff.rc.i_aq_mode == 1                                    # x264's default: variance AQ rides on CRF
    og = O.OracleEncoder(O.default_config(w, h, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, aq_mode=1, aq_strength=1.0397, **eff_kw(eff)))
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.set_qp(rows[i][1], rows[i][1]); og.set_qpm(rows[i][5])          # x264_ratecontrol_mb_qp: (int)(rc->qpm + the AQ offset + 0.5f)
        og.encode(f, 2 if rows[i][0] else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


@pytest.mark.skipif(not __import__("os").path.exists(O.LSMASH_REF), reason="oracle/_ref/liblsmash_ref.so not built")
def test_stream_headers_read_back_by_lsmash(gpu):
    """the stream of x264_encoder_encode (VfW mode: SPS/PPS before every keyframe) as read by L-SMASH, the parser the reference's
    mp4 muxer relies on (oracle/_ref, built from /root/reference/output/L-SMASH): VUI (sar, range, colour description, timing),
    cropping, reference count, CABAC (High profile), and every slice header (cabac_init_idc included: the fields after it must line up)"""
    w, h, nfr = 208, 120, 7
    frames = synth_frames(w, h, nfr, seed=4)
    h_, eff = open_encoder(w, h, {"qp": 26, "keyint": 3, "no-scenecut": None, "ref": 2, "sar": "4:3", "fullrange": "1", "colorprim": "bt709", "transfer": "bt709",
                                  "colormatrix": "bt709", "fps": "30000/1001"}, b"high")
    stream, info, _ = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    sps, pps, sl = O.lsmash_parse(stream)
    assert (sps.profile_idc, sps.cropped_width, sps.cropped_height, sps.max_num_ref_frames, sps.pic_order_cnt_type) == (100, w, h, 2, 2)
    assert (sps.sar_width, sps.sar_height, sps.video_full_range_flag) == (4, 3, 1)
    assert (sps.colour_primaries, sps.transfer_characteristics, sps.matrix_coefficients) == (1, 1, 1)
    assert (sps.num_units_in_tick, sps.time_scale) == (1001, 60000)
    assert (pps.entropy_coding_mode_flag, pps.num_ref_idx_l0_default_active_minus1, pps.deblocking_filter_control_present_flag) == (1, 1, 1)
    assert [(s.nal_unit_type, s.slice_type, s.frame_num) for s in sl] == [(5, 2, 0) if i % 3 == 0 else (1, 0, i % 3) for i in range(nfr)]
    assert [s.idr_pic_id for s in sl if s.idr] == [0, 1, 2]


def test_scenecut_inside_min_keyint_is_a_non_idr_i_picture(gpu):
    """x264_slicetype_decide: a scenecut closer than min-keyint to the last IDR becomes an I picture that keeps the references
    (nal_unit_type 1, slice_type I, frame_num keeps counting); decodes to the encoder's reconstruction"""
    w, h = 176, 144
    frames = synth_frames(w, h, 3, seed=5) + synth_frames(w, h, 3, seed=99)
    h_, eff = open_encoder(w, h, {"qp": 27, "keyint": 250, "min-keyint": 25}, b"high")
    stream, rows, recons = encode_with_decisions(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert [r[0] for r in rows] == [1, 0, 0, 0, 0, 0] and rows[3][2] == 1          # detected, but not a keyframe
    if __import__("os").path.exists(O.LSMASH_REF):
        _, _, sl = O.lsmash_parse(stream)
        assert [(s.nal_unit_type, s.slice_type, s.frame_num) for s in sl] == [(5, 2, 0), (1, 0, 1), (1, 0, 2), (1, 2, 3), (1, 0, 4), (1, 0, 5)]
    dec = O.h264_decode(stream, len(frames), w, h)
    for i in range(len(frames)):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")


def test_abr_steers_towards_the_bitrate(gpu):
    """--bitrate (single-pass ABR, no VBV; the 1-pass branch of x264's rate_estimate_qscale): the quantiser follows the rate
    factor that would have met the target so far, so two targets a factor of four apart give clearly different quantisers and
    sizes, each within a loose band of its target once the controller has settled; the stream stays decodable."""
    w, h, nfr, fps = 352, 288, 50, 25
    frames = synth_frames(w, h, nfr, seed=8, scene_len=10 ** 9)
    res = {}
    for kbps in (400, 1600):
        h_, eff = open_encoder(w, h, {"bitrate": kbps, "keyint": 250, "no-scenecut": None}, b"high")
        assert eff.rc.i_rc_method == HL.X264_RC_ABR and eff.rc.i_bitrate == kbps
        stream, rows, recons = encode_with_decisions(h_, w, h, frames)
        H.x264_encoder_close(h_)
        dec = O.h264_decode(stream, nfr, w, h)
        assert all(np.array_equal(d, r) for d, r in zip(dec, recons))
        res[kbps] = (len(stream) * 8 / (nfr / fps) / 1000, [r[1] for r in rows])
    lo, hi = res[400], res[1600]
    assert 0.5 * 400 < lo[0] < 2.0 * 400 and 0.5 * 1600 < hi[0] < 2.0 * 1600, (lo[0], hi[0])
    assert np.mean(lo[1][10:]) > np.mean(hi[1][10:]) + 4                        # ~4x the bits is ~12 quantiser steps in theory
    assert len(set(lo[1])) > 2                                                  # it actually moves


def test_crf_with_macroblock_tree(gpu, monkeypatch):
    """the driver's default rate control in full: CRF + variance AQ + macroblock-tree over rc-lookahead pictures held back in the
    lookahead queue.  The frame quantiser is the constant crf + 13.5 (1 - qcomp) (the tree does the complexity weighting), the
    macroblock offsets are AQ - tree; the whole chain is replayed with the CPU oracle (lookahead records -> x264o_mbtree -> oracle
    pipeline with those offsets) and must reproduce the host encoder's reconstruction; the stream decodes to it as well."""
    w, h, crf, look = 176, 144, 24.0, 4
    frames = synth_frames(w, h, 7, seed=5) + synth_frames(w, h, 4, seed=99)
    opts = {"crf": crf, "keyint": 250, "min-keyint": 3, "rc-lookahead": look}
    monkeypatch.setenv("X264GPU_HOST_PIPELINE", "0")                         # GPU stage and entropy coding in the same call: the reconstruction read
    h_, eff = open_encoder(w, h, opts, b"high")                              # back after a call is that call's picture
    assert (eff.rc.b_mb_tree, eff.rc.i_lookahead, eff.rc.i_aq_mode) == (1, look, 1)
    stream, rows, recons = encode_with_decisions(h_, w, h, frames)
    H.x264_encoder_close(h_)
    # the pipelined session (the default: the next picture's GPU stage runs behind this one's entropy coding) hands every picture
    # back one call later and writes the same bytes
    monkeypatch.delenv("X264GPU_HOST_PIPELINE")
    hp, _ = open_encoder(w, h, opts, b"high")
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, nn = C.POINTER(HL.Nal)(), C.c_int()
    piped, first_out = b"", None
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
        pic.i_pts = i
        size = H.x264_encoder_encode(hp, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
        assert size >= 0
        if size:
            first_out = i if first_out is None else first_out
            piped += C.string_at(nal[0].p_payload, size)
    while H.x264_encoder_delayed_frames(hp):
        size = H.x264_encoder_encode(hp, C.byref(nal), C.byref(nn), None, C.byref(out))
        assert size > 0
        piped += C.string_at(nal[0].p_payload, size)
    H.x264_encoder_close(hp)
    H.x264_picture_clean(C.byref(pic))
    assert first_out == look + 1 and piped == stream
    n = len(frames)
    types = [2 if r[0] else (1 if r[4] == 2 else 0) for r in rows]            # X264_TYPE_I == 2: I picture that is not IDR
    assert types[0] == 2 and types[7] in (1, 2) and sum(1 for t in types if t) == 2
    frame_qp = int(crf + 13.5 * (1 - 0.6) + 0.5)
    assert all(r[1] == frame_qp for r, t in zip(rows, types) if t == 0)
    dec = O.h264_decode(stream, n, w, h)
    ol = O.OracleLookahead(w, h)
    infos = [ol.frame_cost(f, i == 0)[1] for i, f in enumerate(frames)]
    aqs = [O.aq_offsets(f, w, h, O.AQ1) for f in frames]
    og = O.OracleEncoder(O.default_config(w, h, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, aq_mode=1, aq_strength=1.0397, **eff_kw(eff)))
    bw, bh = (w + 15) // 16, (h + 15) // 16
    lowered = 0
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        last = i                                                            # the window: this picture + the P pictures queued behind it
        while last + 1 < min(n, i + look + 1) and types[last + 1] == 0:
            last += 1
        off = O.mbtree(bw, bh, infos[i:last + 1], aqs[i:last + 1], O.TREE)
        lowered += int((off < aqs[i]).sum())
        og.set_mb_qp_offsets(off)
        og.set_qp(rows[i][1], rows[i][1]); og.set_qpm(rows[i][5])
        og.encode(f, 2 if types[i] == 2 else 3 if types[i] == 1 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle chain picture {i}")
    assert lowered > 50


def test_random_option_mixes_stay_decodable(gpu):
    """host-API stress: random picture sizes (ragged, tiny, wide) x random mixes of the options this round implements x every rate
    control mode, serial and GOP-parallel; each session must return every picture exactly once, in order, and the stream must decode
    to the reconstruction the encoder holds for the last picture (closed loop through the checker decoder)"""
    rng = np.random.default_rng(20261002)
    sizes = [(16, 16), (48, 32), (176, 144), (200, 104), (72, 136), (416, 64), (34, 50), (250, 18)]
    for trial in range(int(__import__("os").environ.get("X264GPU_STRESS_TRIALS", "28"))):
        w, h = sizes[int(rng.integers(len(sizes)))]
        w, h = w & ~1, h & ~1
        nfr = int(rng.integers(2, 12))
        opts = {"keyint": int(rng.choice([1, 2, 3, 5, 250])), "ref": int(rng.integers(1, 5)), "subme": int(rng.integers(0, 8)),
                "me": str(rng.choice(["dia", "hex", "umh", "esa"])), "merange": int(rng.choice([4, 8, 16]))}
        if rng.random() < 0.5: opts["partitions"] = str(rng.choice(["none", "all", "p8x8,i4x4", "i8x8,i4x4", "p8x8"]))
        if rng.random() < 0.3: opts["no-8x8dct"] = None
        if rng.random() < 0.3: opts["no-deblock"] = None
        elif rng.random() < 0.4: opts["deblock"] = f"{int(rng.integers(-3, 4))}:{int(rng.integers(-3, 4))}"
        if rng.random() < 0.3: opts["no-chroma-me"] = None
        if rng.random() < 0.3: opts["no-mixed-refs"] = None
        if rng.random() < 0.3: opts["chroma-qp-offset"] = int(rng.integers(-6, 7))
        if rng.random() < 0.3: opts["no-scenecut"] = None
        mode = str(rng.choice(["qp", "crf", "crf-tree", "abr"]))
        if mode == "qp": opts["qp"] = int(rng.integers(8, 48))
        elif mode == "abr": opts["bitrate"] = int(rng.integers(50, 2000))
        else:
            opts["crf"] = int(rng.integers(12, 40))
            if mode == "crf": opts["no-mbtree"] = None
            else: opts["rc-lookahead"] = int(rng.integers(1, 7))
        if rng.random() < 0.3: opts["aq-mode"] = int(rng.integers(0, 2))
        threads = int(rng.choice([1, 1, 2, 3]))
        if threads > 1:
            opts["threads"] = threads
            opts["keyint"] = min(opts["keyint"], 5)
        opts["min-keyint"] = max(1, min(opts["keyint"], int(rng.integers(1, 4))))
        frames = synth_frames(w, h, nfr, seed=1000 + trial)
        if rng.random() < 0.5:                                               # a cut somewhere
            frames[nfr // 2:] = synth_frames(w, h, nfr - nfr // 2, seed=5000 + trial)
        tag = f"trial {trial}: {w}x{h} x{nfr} {opts}"
        h_, eff = open_encoder(w, h, opts, None if rng.random() < 0.5 else b"high")
        pic, out = HL.Picture(), HL.Picture()
        assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        stream, pts = b"", []
        for i, f in enumerate(frames):
            C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
            pic.i_pts = 7 + 3 * i
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
            assert size >= 0, tag
            if size:
                stream += C.string_at(nal[0].p_payload, size); pts.append(out.i_pts)
            assert H.x264_encoder_delayed_frames(h_) == i + 1 - len(pts), tag
        guard = 0
        while H.x264_encoder_delayed_frames(h_):
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
            assert size > 0, tag
            stream += C.string_at(nal[0].p_payload, size); pts.append(out.i_pts)
            guard += 1
            assert guard <= nfr, tag
        assert pts == [7 + 3 * i for i in range(nfr)], tag
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        got_rec = eff.i_threads <= 1 and H.x264host_get_recon(h_, rec.ctypes.data) == 0
        H.x264_encoder_close(h_)
        H.x264_picture_clean(C.byref(pic))
        dec = O.h264_decode(stream, nfr, w, h)
        assert len(dec) == nfr, tag
        if got_rec:
            np.testing.assert_array_equal(dec[-1], rec, err_msg=tag)


@pytest.mark.parametrize("w,h,opts,slices", [(176, 288, {"qp": 26, "keyint": 5, "no-scenecut": None, "sliced-threads": None, "threads": 3}, 3),
                                            (96, 336, {"qp": 30, "keyint": 250, "no-scenecut": None, "tune-sliced": None}, 5),
                                            (208, 144, {"qp": 24, "keyint": 4, "no-scenecut": None, "sliced-threads": None, "threads": 9}, 2)])
def test_sliced_threads_through_the_api(gpu, w, h, opts, slices):
    """x264's slice threads (--sliced-threads --threads N, what --tune zerolatency switches on): N slices per picture, every slice its own
    wavefront on the device and its own NAL.  Zero delay; the stream decodes to the encoder's reconstruction; the oracle pipeline with the
    same slice count gives the same pictures; more slices than one per four macroblock rows are cut back (x264 validate_parameters)"""
    opts = dict(opts)
    preset_tune = None
    if "tune-sliced" in opts:          # sliced threads from the tune, slice count "auto" = as many as the picture allows
        del opts["tune-sliced"]
        preset_tune = b"zerolatency"
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", preset_tune) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den = 25, 1
    p.i_log_level = -1
    if "bframes" not in opts:
        opts = dict(opts, bframes=0)          # these sessions test the I / P behaviours (one picture back per call); B sessions: test_b_session_*
        opts.setdefault("weightp", 0)         # ... of the ring path (--weightp 2 moves a session without B pictures onto the DPB model: test_weightp_session_*)
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, k
    p.b_vfr_input = 0
    p.b_annexb, p.b_repeat_headers = 1, 1
    assert H.x264_param_apply_profile(C.byref(p), b"high") == 0
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    eff = HL.Param()
    H.x264_encoder_parameters(h_, C.byref(eff))
    assert eff.b_sliced_threads == 1 and eff.i_threads == slices
    nfr = 6
    frames = synth_frames(w, h, nfr, seed=77 + w)
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    for i, (_, _, _, types) in enumerate(info):
        assert sum(1 for t in types if t in (1, 5)) == slices, (i, types)      # one slice NAL per slice, every call returns its picture
    dec = O.h264_decode(stream, nfr, w, h)
    og = O.OracleEncoder(O.default_config(w, h, slices=slices, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=max(0, opts["qp"] - 3), qp_p=opts["qp"], **eff_kw(eff)))
    keyint = opts["keyint"]
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if i % keyint == 0 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


@pytest.mark.parametrize("w,h,opts,slices", [(176, 288, {"qp": 26, "keyint": 5, "no-scenecut": None, "slices": 18}, 18),
                                            (96, 336, {"qp": 30, "keyint": 250, "no-scenecut": None, "slices": 4}, 4),
                                            (208, 144, {"qp": 24, "keyint": 4, "no-scenecut": None, "slices": 40, "no-cabac": None}, 9)])
def test_plain_slices_through_the_api(gpu, w, h, opts, slices):
    """x264's --slices N: N slices per picture (at most one per macroblock row: validate_parameters clips), every slice its own wavefront on the
    device and its own NAL, filtered across the boundaries (disable_deblocking_filter_idc 0).  Zero delay; the stream decodes to the
    encoder's reconstruction; the oracle pipeline with slices_plain gives the same pictures"""
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den = 25, 1
    p.i_log_level = -1
    if "bframes" not in opts:
        opts = dict(opts, bframes=0)          # these sessions test the I / P behaviours (one picture back per call); B sessions: test_b_session_*
        opts.setdefault("weightp", 0)         # ... of the ring path (--weightp 2 moves a session without B pictures onto the DPB model: test_weightp_session_*)
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, k
    p.b_vfr_input = 0
    p.b_annexb, p.b_repeat_headers = 1, 1
    assert H.x264_param_apply_profile(C.byref(p), b"high") == 0
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    eff = HL.Param()
    H.x264_encoder_parameters(h_, C.byref(eff))
    assert eff.b_sliced_threads == 0 and eff.i_slice_count == slices and eff.i_threads == 1
    nfr = 6
    frames = synth_frames(w, h, nfr, seed=78 + w)
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    for i, (_, _, _, types) in enumerate(info):
        assert sum(1 for t in types if t in (1, 5)) == slices, (i, types)
    dec = O.h264_decode(stream, nfr, w, h)
    og = O.OracleEncoder(O.default_config(w, h, slices=slices, slices_plain=1, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=max(0, opts["qp"] - 3), qp_p=opts["qp"], **eff_kw(eff)))
    keyint = opts["keyint"]
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if i % keyint == 0 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


@pytest.mark.parametrize("w,h,opts,okw", [
    (1920, 1080, {"qp": 23, "keyint": 250, "no-scenecut": None}, dict(me_method=1)),                                              # BASELINE.json config 2 (headline size)
    (1280, 720, {"qp": 23, "keyint": 250, "no-scenecut": None}, dict(me_method=1)),                                               # config 1
    (3840, 2160, {"qp": 26, "keyint": 250, "no-scenecut": None, "me": "umh", "ref": 5, "sliced-threads": None}, dict(me_method=2, refs=5)),   # config 3's search (umh, ref 5) at its size
])
def test_full_size_round_trip(gpu, w, h, opts, okw):
    """BASELINE.json's full sizes through the x264 API: the stream (High profile, CABAC) decodes to the encoder's own reconstruction, and the
    oracle pipeline fed the same pictures reconstructs the same samples — at sizes where only a few pictures fit a test's time budget"""
    nfr = 2 if w > 1920 else 3
    h_, eff = open_encoder(w, h, opts, b"high")
    frames = synth_frames(w, h, nfr, seed=w + h)
    stream, info, recons = encode_all(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert eff.b_cabac == 1
    dec = O.h264_decode(stream, nfr, w, h)
    slices = (h + 15) // 16 // 4 if "sliced-threads" in opts else 1
    kw = dict(slices=slices, partitions=7, refs=3, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=opts["qp"] - 3, qp_p=opts["qp"], **eff_kw(eff))
    kw.update(okw)
    og = O.OracleEncoder(O.default_config(w, h, **kw))
    for i, f in enumerate(frames):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"decoded picture {i}")
        og.encode(f, 2 if i == 0 else 0)
        np.testing.assert_array_equal(og.recon(), recons[i], err_msg=f"oracle pipeline picture {i}")


def test_slice_threads_with_gop_slots(gpu, monkeypatch):
    """X264GPU_GOP_SLOTS=G on a slice-threads session: G closed GOPs in lock-step, every picture in slices (slices x G wavefronts of one
    stream) — the bytes equal the zero-delay slice-threads session's, only the delay changes"""
    w, h, nfr, keyint, G = 96, 272, 14, 3, 3
    opts = {"qp": 27, "keyint": keyint, "min-keyint": keyint, "no-scenecut": None, "sliced-threads": None, "threads": 4}
    frames = synth_frames(w, h, nfr, seed=4242)
    h1, e1 = open_encoder(w, h, opts, b"high")
    serial, info1, _ = encode_all(h1, w, h, frames)
    H.x264_encoder_close(h1)
    assert all(sum(1 for t in types if t in (1, 5)) == 4 for _, _, _, types in info1)
    monkeypatch.setenv("X264GPU_GOP_SLOTS", str(G))
    hg, eff = open_encoder(w, h, opts, b"high")
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, n = C.POINTER(HL.Nal)(), C.c_int()
    stream, got = b"", 0
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
        pic.i_pts = i
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size >= 0 and (size == 0 or i >= (G - 1) * keyint)
        if size:
            stream += C.string_at(nal[0].p_payload, size); got += 1
    while H.x264_encoder_delayed_frames(hg):
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        stream += C.string_at(nal[0].p_payload, size); got += 1
    H.x264_encoder_close(hg)
    H.x264_picture_clean(C.byref(pic))
    assert got == nfr and stream == serial


@pytest.mark.parametrize("extra", [{"slices": 9}, {"sliced-threads": None, "threads": 2}])
def test_rate_controlled_session_in_slices(gpu, monkeypatch, extra):
    """the driver's default rate control (CRF + variance AQ + macroblock-tree, pictures held back in the lookahead) on pictures coded in
    slices — --slices N with its repeated slice passes, and slice threads: per-macroblock quantisers cross the slice boundaries (every
    slice starts its delta chain at the slice quantiser), scene cuts put intra macroblocks into P pictures.  The stream decodes to the
    encoder's reconstruction, picture by picture"""
    w, h, look = 176, 144, 3
    frames = synth_frames(w, h, 5, seed=15) + synth_frames(w, h, 4, seed=98)
    monkeypatch.setenv("X264GPU_HOST_PIPELINE", "0")                         # the reconstruction read back after a call is that call's picture
    h_, eff = open_encoder(w, h, dict({"crf": 25.0, "keyint": 250, "min-keyint": 3, "rc-lookahead": look}, **extra), b"high")
    assert (eff.rc.b_mb_tree, eff.rc.i_aq_mode) == (1, 1)
    ns = 9 if "slices" in extra else 2
    stream, rows, recons = encode_with_decisions(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert len(recons) == len(frames)
    nals = [n for n in stream.split(b"\x00\x00\x01") if n and (n[0] & 31) in (1, 5)]
    assert len(nals) == ns * len(frames)
    dec = O.h264_decode(stream, len(frames), w, h)
    for i in range(len(frames)):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"picture {i}")


def test_plain_slices_with_gop_slots(gpu):
    """--slices N --threads G: G closed GOPs in lock-step with every picture in N slices (N x G wavefronts of one stream; the slots keep their own
    intra statistics for the repeated slice passes) — the bytes equal the --threads 1 session's, only the delay changes"""
    w, h, nfr, keyint, G = 96, 272, 14, 3, 3
    opts = {"qp": 27, "keyint": keyint, "min-keyint": keyint, "no-scenecut": None, "slices": 17}
    frames = synth_frames(w, h, nfr, seed=4243, scene_len=3)
    h1, e1 = open_encoder(w, h, opts, b"high")
    serial, info1, _ = encode_all(h1, w, h, frames)
    H.x264_encoder_close(h1)
    assert all(sum(1 for t in types if t in (1, 5)) == 17 for _, _, _, types in info1)
    hg, eff = open_encoder(w, h, dict(opts, threads=G), b"high")
    assert eff.i_threads == G and eff.i_slice_count == 17 and eff.b_sliced_threads == 0
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, n = C.POINTER(HL.Nal)(), C.c_int()
    stream, got = b"", 0
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
        pic.i_pts = i
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size >= 0 and (size == 0 or i >= (G - 1) * keyint)
        if size:
            stream += C.string_at(nal[0].p_payload, size); got += 1
    while H.x264_encoder_delayed_frames(hg):
        size = H.x264_encoder_encode(hg, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        stream += C.string_at(nal[0].p_payload, size); got += 1
    H.x264_encoder_close(hg)
    H.x264_picture_clean(C.byref(pic))
    assert got == nfr and stream == serial


def test_access_unit_delimiters(gpu):
    """--aud: every access unit starts with a type-9 NAL whose primary_pic_type tells I from P (7.3.2.4); the rest of the stream is unchanged"""
    w, h, nfr = 96, 80, 5
    frames = synth_frames(w, h, nfr, seed=9)
    opts = {"qp": 28, "keyint": 3, "no-scenecut": None}
    h0, _ = open_encoder(w, h, opts, b"high")
    plain, _, _ = encode_all(h0, w, h, frames)
    H.x264_encoder_close(h0)
    h1, eff = open_encoder(w, h, dict(opts, aud=None), b"high")
    assert eff.b_aud == 1
    stream, info, recons = encode_all(h1, w, h, frames)
    H.x264_encoder_close(h1)
    for i, (_, _, _, types) in enumerate(info):
        assert types[0] == 9 and types.count(9) == 1, (i, types)
    auds = [stream[k + 4:k + 6] for k in range(len(stream) - 6) if stream[k:k + 5] == b"\x00\x00\x00\x01\x09"]
    assert [a[1] >> 5 for a in auds] == [0 if i % 3 == 0 else 1 for i in range(nfr)]          # primary_pic_type
    short = lambda b: b.replace(b"\x00\x00\x00\x01", b"\x00\x00\x01")          # only the first NAL of an access unit has the long start code
    assert short(stream).replace(b"\x00\x00\x01\x09\x10", b"").replace(b"\x00\x00\x01\x09\x30", b"") == short(plain)
    dec = O.h264_decode(stream, nfr, w, h)
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i])


def encode_delayed(h_, w, h, frames):
    """feeds every frame, then drains (codec.c:1842-1856): -> (stream, [(type, keyframe, pts, dts, size)] as the pictures leave)"""
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    stream, recs = b"", []
    nal, n = C.POINTER(HL.Nal)(), C.c_int()

    def take(size):
        nonlocal stream
        assert size >= 0
        if size:
            assert sum(nal[k].i_payload for k in range(n.value)) == size
            stream += C.string_at(nal[0].p_payload, size)
            recs.append((out.i_type, out.b_keyframe, out.i_pts, out.i_dts, size))
    for i, f in enumerate(frames):
        C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
        pic.i_pts = i
        take(H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out)))
    while H.x264_encoder_delayed_frames(h_):
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        take(size)
    H.x264_picture_clean(C.byref(pic))
    return stream, recs


@pytest.mark.parametrize("w,h,n,opts,pattern", [
    (176, 144, 14, {"qp": 23, "keyint": 30, "scenecut": 0, "bframes": 3, "b-adapt": 0}, "IPRBBPRBBPRBBP"),                         # preset medium as the device runs it: bframes 3, b-pyramid, weightb, ref 3
    (128, 96, 12, {"qp": 26, "keyint": 30, "scenecut": 0, "bframes": 1, "ref": 1, "b-adapt": 0}, "IPBPBPBPBPBP"),
    (96, 80, 13, {"crf": 24, "keyint": 6, "min-keyint": 6, "scenecut": 0, "bframes": 2, "b-pyramid": "none", "no-mbtree": None, "b-adapt": 0}, None),
    (176, 144, 20, {"crf": 24, "keyint": 30, "bframes": 3, "rc-lookahead": 8}, None),                               # the driver's default rate control: CRF + AQ + macroblock-tree through B pictures
    (176, 144, 16, {"qp": 23, "keyint": 30, "bframes": 3}, None),                                                                  # medium's lookahead: --b-adapt 1 --scenecut 40 on the device's (p0, p1, b) frame costs
    (176, 144, 9, {"qp": 23, "keyint": 30, "scenecut": 0, "bframes": 0, "weightp": 2}, "IPPPPPPPP"),                # no B pictures, --weightp 2: the DPB model with no delay
    (176, 288, 14, {"crf": 24, "keyint": 30, "bframes": 3, "rc-lookahead": 6, "slices": 6}, None),                  # B pictures in --slices 6
    (176, 288, 14, {"crf": 24, "keyint": 30, "bframes": 3, "rc-lookahead": 6, "sliced-threads": None, "threads": 3}, None),      # ... and in x264's slice threads
    (176, 144, 30, {"bitrate": 600, "keyint": 30, "bframes": 3, "rc-lookahead": 10}, None),
    (176, 144, 22, {"crf": 24, "keyint": 60, "bframes": 3, "b-adapt": 2, "rc-lookahead": 14}, None),                # presets slower and up: the trellis over picture types on the device's frame costs                          # single-pass ABR keeps B pictures, the lookahead and macroblock-tree
])
def test_b_session_through_the_encode_api(gpu, w, h, n, opts, pattern):
    """B pictures through x264_encoder_encode (codec.c:1693): types / pts / dts as x264 hands them to the muxers (output/matroska.c:199-202), the
    stream decodes to every source picture, and equals the checker's stream for the same schedule byte for byte"""
    frames = synth_frames(w, h, n, seed=4)
    h_, eff = open_encoder(w, h, opts, profile=None)
    assert eff.i_bframe == opts.get("bframes", 3) and (not eff.i_bframe or (eff.i_bframe_adaptive == opts.get("b-adapt", 1) and eff.analyse.i_direct_mv_pred == 1))
    assert eff.analyse.i_weighted_pred == (2 if opts.get("ref", 3) >= 2 else 0)          # medium's default, kept: the blind duplicate of reference 0 (it needs two references)
    stream, recs = encode_delayed(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert len(recs) == n
    TYPE = {1: "I", 2: "i", 3: "P", 4: "R", 5: "B"}
    got = "".join(TYPE[r[0]] for r in recs)
    if pattern:
        assert got == pattern
    dts = [r[3] for r in recs]
    assert dts == sorted(dts) and len(set(dts)) == n and all(r[3] <= r[2] for r in recs)
    dec = O.h264_decode(stream, n, w, h)
    pocs = O.h264_last_pocs()
    assert len(dec) == n
    # POC restarts at every IDR: display index = pts
    for d, r in zip(dec, recs):
        assert psnr(d[:w * h], frames[r[2]][:w * h]) > 30.0
    assert [p for p in pocs] == [2 * (r[2] - max(q[2] for q in recs if q[1] and q[2] <= r[2])) for r in recs]


def test_random_b_session_mixes_stay_decodable(gpu):
    """host-API stress for the sessions on the DPB model: random picture sizes x random mixes of B-picture structure (bframes, b-adapt 0 / 1 / 2,
    b-pyramid, weightb, weightp 0 / 1 / 2), references, searches, partitions (p8x8 / b8x8 apart), trellis, slices / slice threads, scene cuts, short
    GOPs and every rate control mode that runs there (CQP, CRF with and without macroblock-tree, single-pass ABR, AQ on / off).  Every picture must
    come back exactly once with a monotone dts, the stream must decode to as many pictures, and every decoded picture must resemble its source"""
    rng = np.random.default_rng(20261003)
    sizes = [(64, 48), (176, 144), (200, 104), (72, 136), (128, 96), (96, 80), (160, 128)]
    for trial in range(int(__import__("os").environ.get("X264GPU_STRESS_TRIALS", "24"))):
        w, h = sizes[int(rng.integers(len(sizes)))]
        nfr = int(rng.integers(3, 20))
        opts = {"bframes": int(rng.integers(0, 4)), "b-adapt": int(rng.integers(0, 3)), "ref": int(rng.integers(1, 5)),
                "me": str(rng.choice(["dia", "hex", "umh", "esa"])), "merange": int(rng.choice([4, 8, 16])), "subme": int(rng.choice([7, 7, 7, 8, 9, 6])),
                "keyint": int(rng.choice([2, 4, 9, 250])), "weightp": int(rng.integers(0, 3))}
        if rng.random() < 0.3: opts["b-pyramid"] = str(rng.choice(["none", "normal", "strict"]))
        if rng.random() < 0.3: opts["no-weightb"] = None
        if rng.random() < 0.4: opts["partitions"] = str(rng.choice(["none", "all", "p8x8,i4x4", "b8x8,i8x8,i4x4", "p8x8,b8x8", "i8x8,i4x4"]))
        if rng.random() < 0.3: opts["no-8x8dct"] = None
        if rng.random() < 0.3: opts["no-deblock"] = None
        if rng.random() < 0.3: opts["no-mixed-refs"] = None
        if rng.random() < 0.3: opts["trellis"] = int(rng.integers(0, 3))
        if rng.random() < 0.3: opts["no-psy"] = None
        if rng.random() < 0.3: opts["direct"] = str(rng.choice(["spatial", "auto", "temporal"]))
        if rng.random() < 0.3: opts["scenecut"] = int(rng.choice([0, 40, 80]))
        if rng.random() < 0.25: opts["slices"] = int(rng.integers(2, 6))
        elif rng.random() < 0.2 and h >= 128: opts.update({"sliced-threads": None, "threads": 2})
        mode = str(rng.choice(["qp", "crf", "crf-tree", "abr"]))
        if mode == "qp": opts["qp"] = int(rng.integers(14, 40))
        elif mode == "abr": opts["bitrate"] = int(rng.integers(100, 1500)); opts["rc-lookahead"] = int(rng.integers(0, 12))
        else:
            opts["crf"] = int(rng.integers(16, 36))
            if mode == "crf": opts["no-mbtree"] = None
            else: opts["rc-lookahead"] = int(rng.integers(1, 12))
        if rng.random() < 0.3: opts["aq-mode"] = int(rng.integers(0, 2))
        opts["min-keyint"] = max(1, min(opts["keyint"], int(rng.integers(1, 4))))
        frames = synth_frames(w, h, nfr, seed=3000 + trial)
        if rng.random() < 0.4:                                               # a cut somewhere
            frames[nfr // 2:] = synth_frames(w, h, nfr - nfr // 2, seed=7000 + trial)
        tag = f"trial {trial}: {w}x{h} x{nfr} {opts}"
        h_, eff = open_encoder(w, h, opts, profile=None)
        stream, recs = encode_delayed(h_, w, h, frames)
        H.x264_encoder_close(h_)
        assert sorted(r[2] for r in recs) == list(range(nfr)), tag
        dts = [r[3] for r in recs]
        assert dts == sorted(dts) and all(r[3] <= r[2] for r in recs), tag
        if not eff.i_bframe:
            assert [r[2] for r in recs] == list(range(nfr)), tag
        dec = O.h264_decode(stream, nfr, w, h)
        assert len(dec) == nfr, tag
        for d, r in zip(dec, recs):
            assert psnr(d[:w * h], frames[r[2]][:w * h]) > 16.0, (tag, r)          # (noise content at quantisers up to 40: a floor that only garbage falls under)


@pytest.mark.parametrize("preset", ["ultrafast", "superfast", "veryfast", "faster", "fast", "medium", "slow", "slower", "veryslow", "placebo"])
def test_every_preset_codes_a_decodable_stream(gpu, preset):
    """x264_param_default_preset(name) as the driver calls it (codec.c:1463) for each of x264's ten presets, rate control left at the driver's default
    (CRF 23): the session opens, reports what it runs (B pictures from superfast up — below subme 7 their slices are analysed without RD; bframes 8 / 16 and
    --b-adapt 2 in veryslow / placebo), returns every picture once and the stream decodes to the source"""
    w, h, n = 176, 144, 36
    frames = synth_frames(w, h, n, seed=31, scene_len=23)
    h_, eff = open_encoder(w, h, {"bframes": {"ultrafast": 0, "veryslow": 8, "placebo": 16}.get(preset, 3), "weightp": {"ultrafast": 0, "superfast": 1, "veryfast": 1, "faster": 1, "fast": 1}.get(preset, 2)},
                           profile=None, preset=preset.encode())
    if preset != "ultrafast":
        assert eff.i_bframe == {"veryslow": 8, "placebo": 16}.get(preset, 3) and eff.i_bframe_adaptive == (2 if preset in ("slower", "veryslow", "placebo") else 1)
    else:
        assert eff.i_bframe == 0
    # RD refinement (subme 8) runs from slow up (umh); placebo's tesa maps to esa, where it does not (subme 7)
    assert eff.analyse.i_subpel_refine == {"ultrafast": 0, "superfast": 1, "veryfast": 2, "faster": 4, "fast": 6, "medium": 7, "slow": 8, "placebo": 7}.get(preset, 9), preset          # (slower 9; veryslow 10 -> 9; placebo's tesa -> esa keeps subme 7)
    # ... and the rest of the preset's list, or exactly the documented downgrade (tests/test_effective_params_cpu.py has the table and checks the log lines):
    # ref > 5 -> 5, tesa -> esa, p4x4 off, trellis as the preset says wherever CABAC + subme >= 6 hold
    X264_ANALYSE_PSUB8x8 = 0x0020
    assert eff.i_frame_reference == {"ultrafast": 1, "superfast": 1, "veryfast": 1, "faster": 2, "fast": 2, "medium": 3}.get(preset, 5), preset
    assert eff.analyse.i_me_method == {"ultrafast": 0, "superfast": 0, "slower": 2, "veryslow": 2, "placebo": 3}.get(preset, 1), preset          # dia / hex / umh / esa
    assert eff.analyse.i_trellis == {"ultrafast": 0, "superfast": 0, "veryfast": 0, "faster": 0, "fast": 1, "medium": 1}.get(preset, 2), preset      # (faster: trellis 1 needs subme >= 6 here)
    assert not (eff.analyse.inter & X264_ANALYSE_PSUB8x8), preset
    stream, recs = encode_delayed(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert sorted(r[2] for r in recs) == list(range(n))
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n
    for d, r in zip(dec, recs):
        assert psnr(d[:w * h], frames[r[2]][:w * h]) > 26.0, (preset, r)


@pytest.mark.parametrize("opts", [
    {"subme": 8, "me": "umh", "trellis": 2, "ref": 5, "bframes": 3, "b-adapt": 2, "rc-lookahead": 10, "keyint": 12, "qp": 24, "direct": "auto"},          # preset slow's toolset
    {"subme": 7, "me": "hex", "trellis": 1, "ref": 3, "bframes": 3, "b-adapt": 0, "keyint": 16, "qp": 25, "direct": "temporal"},
    {"subme": 8, "me": "hex", "trellis": 1, "ref": 2, "bframes": 0, "keyint": 9, "crf": 25, "rc-lookahead": 4},
    {"subme": 6, "me": "hex", "trellis": 1, "ref": 2, "bframes": 3, "b-adapt": 1, "rc-lookahead": 8, "keyint": 12, "crf": 24, "mixed-refs": 0},      # preset fast: B slices without RD, trellis in their final encode
    {"subme": 4, "me": "hex", "trellis": 0, "ref": 2, "bframes": 3, "b-adapt": 1, "rc-lookahead": 6, "keyint": 12, "qp": 25, "mixed-refs": 0},       # preset faster
    {"subme": 5, "me": "hex", "trellis": 0, "ref": 2, "bframes": 3, "b-adapt": 1, "rc-lookahead": 6, "keyint": 12, "qp": 25, "no-cabac": None},      # Main profile CAVLC with B pictures
    {"subme": 2, "me": "hex", "trellis": 0, "ref": 1, "bframes": 3, "b-adapt": 1, "rc-lookahead": 4, "keyint": 10, "qp": 26, "mixed-refs": 0},
    {"subme": 7, "me": "hex", "ref": 3, "bframes": 3, "b-adapt": 1, "rc-lookahead": 6, "keyint": 12, "crf": 25, "no-cabac": None},      # medium --no-cabac keeps its B pictures: RD on CAVLC bit counts
    {"subme": 9, "me": "umh", "trellis": 2, "ref": 4, "bframes": 3, "b-adapt": 2, "rc-lookahead": 8, "keyint": 12, "qp": 24, "direct": "auto"},          # preset slower's analysis: subme 9
])
def test_rd_refinement_session_equals_the_checker(gpu, tmp_path, opts):
    """a --subme 8 session (RD refinement of the P partitions' vectors and of the intra modes, B slices one level down) through x264_encoder_encode on
    the device and over the CPU checker: same picture types, timestamps and bytes"""
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub")])
    _session_equals_checker(tmp_path, 176, 144, 14, opts, 77, 9, "moving")


def _session_equals_checker(tmp_path, w, h, nfr, opts, seed, scene, kind):
    """one session over the device (in process) and over the CPU checker (tests/stub, a child process): same picture types, pts, dts and bytes"""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "stub"))
    from run_host_b import make_frames
    frames = make_frames(w, h, nfr, seed, scene_len=scene, static=int(kind == "static"), fade=4 if kind == "fade" else 0)
    tag = f"{w}x{h} x{nfr} {opts} seed {seed} scene_len {scene} {kind}"
    h_, eff = open_encoder(w, h, opts, profile=None)
    stream, recs = encode_delayed(h_, w, h, frames)
    H.x264_encoder_close(h_)
    out = str(tmp_path / "chk.h264")
    args = [f"{k}={v}" if v is not None else k for k, v in opts.items()] + ([f"scene_len={scene}"] if scene else []) + (["static=1"] if kind == "static" else ["fade=4"] if kind == "fade" else [])
    r = subprocess.run([sys.executable, os.path.join(here, "stub", "run_host_b.py"), out, str(w), str(h), str(nfr), str(seed)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (tag, r.stderr[-1500:])
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert [(x[0], x[1], x[2]) for x in info["recs"]] == [(t, pts, dts) for t, _, pts, dts, _ in recs], tag
    assert open(out, "rb").read() == stream, tag


@pytest.mark.parametrize("w,h,nfr,seed,scene,kind,opts", [
    # (bframes / weightp named because this file's open_encoder switches them off otherwise: they are medium's own values)
    (176, 144, 60, 5, 23, "moving", {"crf": 23, "keyint": 250, "rc-lookahead": 40, "bframes": 3, "weightp": 2}),          # = tests/test_decisions_cpu.py DEFAULT_CASES[1]
    (208, 112, 36, 7, 0, "fade", {"crf": 21, "keyint": 30, "rc-lookahead": 12, "bframes": 3, "weightp": 2}),               # ... [2]: lookahead weights on a fade, the keyint limit
    (176, 144, 30, 9, 11, "fade", {"crf": 24, "keyint": 250, "rc-lookahead": 8, "bframes": 3, "weightp": 0}),            # ... [3]: X264_WEIGHTP_FAKE: the tree's weightdelta
])
def test_default_sessions_equal_the_checker_sessions(gpu, tmp_path, w, h, nfr, seed, scene, kind, opts):
    """the driver's DEFAULT session (codec.c:1504-1507, config.c:109-111: CRF 23 + AQ mode 1 + macroblock-tree over rc-lookahead 40 + b-adapt 1 + weightp 2) on
    the device against the same host code over the CPU checker: same picture types, timestamps and bytes — i.e. the same float quantisers and per-macroblock
    quantisers, which tests/test_decisions_cpu.py test_default_session_equals_the_twin compares float for float with the decision twin on the same clips"""
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub")])
    _session_equals_checker(tmp_path, w, h, nfr, opts, seed, scene, kind)


def test_random_b_sessions_equal_the_checker_sessions(gpu, tmp_path):
    """the same host code over the device and over the CPU checker (tests/stub: the oracle behind the device ABI, in a child process) must write
    the same bytes for the same session: every device primitive a session touches — lookahead frame costs of (p0, p1, b) triples, weight analysis,
    macroblock-tree propagation, AQ, the macroblock loop of I / P / B pictures — is compared through everything the host decides from it
    (picture types, quantisers, weights).  Random option mixes around the driver's defaults, on plain, fading and static clips"""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    subprocess.check_call(["make", "-s", "-C", os.path.join(here, "stub")])
    rng = np.random.default_rng(int(os.environ.get("X264GPU_STRESS_SEED", "20261004")))
    for trial in range(int(os.environ.get("X264GPU_STRESS_TRIALS", "10"))):
        w, h = [(176, 144), (128, 96), (96, 80), (208, 112)][int(rng.integers(4))]
        nfr = int(rng.integers(8, 22))
        opts = {"bframes": int(rng.integers(1, 4)), "b-adapt": int(rng.integers(0, 3)), "ref": int(rng.integers(1, 4)), "keyint": int(rng.choice([6, 12, 250])),
                "weightp": int(rng.integers(0, 3)), "me": str(rng.choice(["dia", "hex", "umh"]))}
        mode = str(rng.choice(["qp", "crf", "crf-tree", "abr"]))
        if mode == "qp": opts["qp"] = int(rng.integers(18, 34))
        elif mode == "abr": opts["bitrate"] = int(rng.integers(150, 900)); opts["rc-lookahead"] = int(rng.integers(2, 10))
        else:
            opts["crf"] = int(rng.integers(18, 32))
            if mode == "crf": opts["no-mbtree"] = None
            else: opts["rc-lookahead"] = int(rng.integers(2, 10))
        if rng.random() < 0.3: opts["slices"] = int(rng.integers(2, 5))
        if rng.random() < 0.3: opts["trellis"] = int(rng.integers(0, 3))
        if rng.random() < 0.3: opts["scenecut"] = 0
        seed = 400 + trial
        scene = int(rng.choice([0, 7]))
        kind = str(rng.choice(["plain", "plain", "fade", "static"]))
        if os.environ.get("X264GPU_STRESS_WIDE"):          # a wider mix for soaks (drawn after the draws above: the default trials stay what they were)
            if rng.random() < 0.3: opts["partitions"] = str(rng.choice(["none", "p8x8,i4x4", "b8x8,i8x8,i4x4", "p8x8,b8x8", "i8x8,i4x4"]))
            if rng.random() < 0.2: opts["no-8x8dct"] = None
            if rng.random() < 0.2: opts["no-deblock"] = None
            if rng.random() < 0.2: opts["no-mixed-refs"] = None
            if rng.random() < 0.2: opts["no-weightb"] = None
            if rng.random() < 0.2: opts["no-psy"] = None
            if rng.random() < 0.2: opts["no-fast-pskip"] = None
            if rng.random() < 0.2: opts["b-pyramid"] = "none"
            if rng.random() < 0.2: opts["aq-mode"] = 0
            if rng.random() < 0.2: opts["merange"] = int(rng.choice([4, 8, 24]))
            if rng.random() < 0.2: opts["chroma-qp-offset"] = int(rng.integers(-4, 5))
            if rng.random() < 0.2: opts["ref"] = int(rng.integers(4, 6))
            if rng.random() < 0.15: opts["bframes"] = int(rng.integers(4, 7))
            if rng.random() < 0.2: opts["min-keyint"] = int(rng.integers(1, 5))
            if rng.random() < 0.15: opts.update({"sliced-threads": None, "threads": 2}); opts.pop("slices", None)
            # round 4's options (drawn last)
            if rng.random() < 0.5: opts["subme"] = int(rng.choice([1, 2, 4, 5, 6, 7, 8, 9, 9]))
            if rng.random() < 0.4: opts["direct"] = str(rng.choice(["spatial", "temporal", "auto"]))
            if rng.random() < 0.2 and opts.get("aq-mode", 1): opts["aq-mode"] = int(rng.choice([2, 3]))
            if rng.random() < 0.15 and opts.get("subme", 7) < 8: opts["no-cabac"] = None
        _session_equals_checker(tmp_path, w, h, nfr, opts, seed, scene, kind)


def test_direct_vector_beyond_the_padding_is_pulled_back(gpu, tmp_path):
    """trial 291 of a 300-session soak: a spatial-direct vector is a neighbour's, taken as it is — in the bottom macroblock row it pointed 34 samples
    below the picture, past the replicated border of the reference planes.  x264's mb_mc clips vectors to mv_min / mv_max before the fetch (same
    samples inside the border); oracle, device and checker decoder now do too"""
    import os
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.abspath(__file__)), "stub")])
    _session_equals_checker(tmp_path, 96, 80, 10, {"bframes": 3, "b-adapt": 2, "ref": 1, "keyint": 6, "weightp": 2, "me": "hex", "bitrate": 554, "rc-lookahead": 3}, 691, 0, "fade")


def test_cross_session_batcher_on_the_device(gpu):
    """X264GPU_BATCH=16: sixteen sessions from sixteen host threads, one lock-step launch per picture on the device; byte-identical to sixteen
    sessions run one after the other"""
    from test_bframes_cpu import _batch
    r = _batch(16, 176, 144, 9, ["qp=23", "keyint=30", "scenecut=0", "b-adapt=0", "bframes=3"], gpu=True)
    assert r["equal"] == [True] * 16 and r["distinct"] == 16, r


def test_cross_session_batcher_with_crf_on_the_device(gpu):
    """CRF sessions in one batch: every stream carries its own float quantiser (x264gpu_pic.qp and its float quantiser qpm) through one
    lock-step launch; byte-identical to the sessions run one by one (round-3 advisor finding: the device used to refuse differing fractions)"""
    from test_bframes_cpu import _batch
    r = _batch(6, 176, 144, 9, ["crf=24", "keyint=8", "min-keyint=8", "scenecut=0", "b-adapt=0", "bframes=2", "no-mbtree"], gpu=True)
    assert r["equal"] == [True] * 6 and r["distinct"] == 6, r


def test_fade_session_gets_luma_weights_on_the_device(gpu):
    """a fade through x264_encoder_encode with medium's lookahead on the device: x264_weights_analyse's restatement (host) on the device's
    statistics and weight costs gives the P pictures luma weights; the stream is smaller than with --weightp 0 and decodes to the source"""
    from test_bframes_cpu import fade_frames
    w, h, n = 176, 144, 14
    frames = fade_frames(w, h, n, 3)
    out = {}
    for wp in (2, 0):
        h_, eff = open_encoder(w, h, {"qp": 23, "keyint": 60, "bframes": 3, "weightp": wp}, profile=None)
        assert eff.analyse.i_weighted_pred == wp
        stream, recs = encode_delayed(h_, w, h, frames)
        H.x264_encoder_close(h_)
        out[wp] = (stream, recs)
    assert len(out[2][0]) < 0.97 * len(out[0][0]), (len(out[2][0]), len(out[0][0]))
    dec = O.h264_decode(out[2][0], n, w, h)
    for d, r in zip(dec, out[2][1]):
        assert psnr(d[:w * h], frames[r[2]][:w * h]) > 33.0


def test_long_default_session_on_the_device(gpu):
    """the driver's default session shape over several GOPs on the device: CRF + AQ + macroblock-tree over a 12-picture lookahead, b-adapt 1,
    scene cuts, weightp 2, keyint 25 — every picture comes back once, dts is monotone, the stream decodes to the source (the queue rings, the DPB
    and the lookahead's slots wrap many times)"""
    w, h, n = 176, 144, 70
    frames = synth_frames(w, h, n, seed=21, scene_len=19)
    h_, eff = open_encoder(w, h, {"crf": 24, "keyint": 25, "min-keyint": 3, "bframes": 3, "rc-lookahead": 12}, profile=None)
    assert (eff.i_bframe, eff.i_bframe_adaptive, eff.rc.b_mb_tree, eff.analyse.i_weighted_pred) == (3, 1, 1, 2)
    stream, recs = encode_delayed(h_, w, h, frames)
    H.x264_encoder_close(h_)
    assert sorted(r[2] for r in recs) == list(range(n))
    dts = [r[3] for r in recs]
    assert dts == sorted(dts) and all(r[3] <= r[2] for r in recs)
    dec = O.h264_decode(stream, n, w, h)
    assert len(dec) == n
    for d, r in zip(dec, recs):
        assert psnr(d[:w * h], frames[r[2]][:w * h]) > 28.0
    assert sum(1 for r in recs if r[1]) >= 3          # IDR pictures: keyint 25 and the scene cuts


def test_third_party_decoder_agrees(gpu):
    """a preset-medium stream (B pictures, CABAC, 8x8 transform) decoded by a decoder found on the box at run time equals the
    encoder's own reconstruction picture for picture.  The harness itself (pts <-> reconstruction, display order) is checked against the
    builder's decoder first, so that a mismatch reported by a third-party decoder means the stream; skips when the box has no such decoder."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import decoder_probe
    w, h, nfr = 176, 144, 9
    stream, recons = decoder_probe.encode_session(w, h, nfr)
    dec = O.h264_decode(stream, nfr, w, h)                # (the builder's decoder returns pictures in decoding order = the order the calls returned them)
    assert sorted(recons) == list(range(nfr)) and list(recons) != sorted(recons), "the session codes B pictures: output order differs from display order"
    for i, pts in enumerate(recons):
        np.testing.assert_array_equal(dec[i], recons[pts], err_msg=f"oracle/h264dec picture {i} (pts {pts})")
    res = decoder_probe.probe(w, h, nfr)
    assert set(res) >= {"found", "decoder", "equal", "pictures", "seen_not_driven"}
    if not res["found"]:
        pytest.skip("no third-party H.264 decoder on this box (ffmpeg / gst-launch-1.0 / rocDecode sample): nothing to compare with; seen: %r" % (res["seen_not_driven"],))
    assert res["equal"] is True, res


@pytest.mark.parametrize("w,h,nfr,keyint,threads,extra", [(176, 144, 23, 8, 3, {}), (96, 80, 21, 8, 2, {"weightp": 2, "b-pyramid": "none", "bframes": 2}), (64, 48, 5, 12, 2, {"direct": "temporal"})])
def test_gop_slots_with_b_pictures_equal_serial(gpu, w, h, nfr, keyint, threads, extra):
    """--threads G with medium's B pictures: closed GOPs of one stream in lock-step on the DPB model — the stream, the picture types and the
    pts / dts of every output equal the threads-1 session's; the last, shorter GOP is coded alone at the flush.  (CPU twin on the stub over several
    devices: tests/test_shard_cpu.py::test_gop_slots_with_b_pictures_equal_the_serial_stream.)"""
    frames = synth_frames(w, h, nfr, seed=17 * w + nfr)
    opts = dict({"qp": 26, "keyint": keyint, "min-keyint": keyint, "no-scenecut": None, "bframes": 3, "b-adapt": 0, "weightp": 0}, **extra)

    def run(thr):
        h_, eff = open_encoder(w, h, dict(opts, threads=thr), b"high")
        assert eff.i_threads == thr and eff.i_bframe == opts["bframes"] and eff.i_bframe_adaptive == 0
        pic, out = HL.Picture(), HL.Picture()
        assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        stream, meta = b"", []

        def take(size):
            nonlocal stream
            if size > 0:
                stream += C.string_at(nal[0].p_payload, size)
                meta.append((int(out.i_type), int(out.b_keyframe), int(out.i_pts), int(out.i_dts), [(int(nal[k].i_type), int(nal[k].i_ref_idc)) for k in range(n.value)]))
        for i, f in enumerate(frames):
            C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
            pic.i_pts = 10 + i
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
            assert size >= 0
            take(size)
        while H.x264_encoder_delayed_frames(h_):
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
            assert size > 0
            take(size)
        H.x264_encoder_close(h_)
        H.x264_picture_clean(C.byref(pic))
        return stream, meta

    serial, m1 = run(1)
    par, mg = run(threads)
    assert len(m1) == len(mg) == nfr and [m[2] for m in m1] != sorted(m[2] for m in m1), "B pictures: output order differs from display order"
    assert mg == m1
    assert par == serial
    dec = O.h264_decode(par, nfr, w, h)
    assert len(dec) == nfr


@pytest.mark.parametrize("w,h,nfr,opts", [
    (352, 288, 26, {"qp": 26, "bframes": 3, "keyint": 12, "no-scenecut": None, "b-adapt": 0}),                           # medium as it is: P, Bref, b, b of a mini-GOP + the next P in flight
    (176, 144, 31, {"crf": 23, "bframes": 3, "rc-lookahead": 10}),                                                    # the driver's default rate control: tree offsets per picture, read while later decisions run
    (176, 144, 22, {"qp": 25, "bframes": 2, "b-pyramid": "none", "ref": 3, "direct": "temporal", "b-adapt": 0}),      # temporal direct: the co-located picture's vectors come from a picture in flight
    (128, 96, 25, {"crf": 24, "bframes": 3, "b-adapt": 2, "weightp": 2, "rc-lookahead": 6, "slices": 2, "keyint": 9, "min-keyint": 3}),
    (64, 48, 3, {"qp": 27, "bframes": 3}),                                                                            # fewer pictures than launch contexts
    (1280, 720, 22, {"crf": 23, "bframes": 3}),                                                                       # the driver's default session at a size where the pictures overlap for real (0.3 - 0.5 s each)
])
def test_pictures_in_flight_equal_serial(gpu, w, h, nfr, opts):
    """Several pictures of ONE session in flight: launch contexts over the shared DPB on streams of their own, each picture behind the events
    of the pictures it references.  The stream, the picture types, pts / dts and nal_ref_idc equal those of the session that codes one picture a call (X264GPU_INFLIGHT=0),
    and the reconstruction of the last picture handed back is the decoder's.  (CPU twin on the stub: tests/test_shard_cpu.py::test_pictures_in_flight_equal_the_serial_stream.)"""
    frames = synth_frames(w, h, nfr, seed=3 * w + nfr)

    def run(inflight):
        old = os.environ.get("X264GPU_INFLIGHT")
        os.environ["X264GPU_INFLIGHT"] = str(inflight)
        try:
            h_, eff = open_encoder(w, h, dict(opts, threads=1), b"high")
        finally:
            if old is None: os.environ.pop("X264GPU_INFLIGHT", None)
            else: os.environ["X264GPU_INFLIGHT"] = old
        assert eff.i_bframe == opts["bframes"]
        assert H.x264host_pictures_in_flight(h_) == (inflight if inflight > 1 else 1), "the path under test: that many launch contexts"
        pic, out = HL.Picture(), HL.Picture()
        assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
        nal, n = C.POINTER(HL.Nal)(), C.c_int()
        stream, meta, first = b"", [], None

        def take(size, i):
            nonlocal stream, first
            if size > 0:
                if first is None: first = i
                stream += C.string_at(nal[0].p_payload, size)
                meta.append((int(out.i_type), int(out.b_keyframe), int(out.i_pts), int(out.i_dts), [(int(nal[k].i_type), int(nal[k].i_ref_idc)) for k in range(n.value)]))
        for i, f in enumerate(frames):
            C.memmove(pic.img.plane[0], f.ctypes.data, f.size)
            pic.i_pts = i
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
            assert size >= 0
            take(size, i)
        while H.x264_encoder_delayed_frames(h_):
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
            assert size > 0
            take(size, nfr)
        rec = np.zeros(w * h * 3 // 2, np.uint8)
        assert H.x264host_get_recon(h_, rec.ctypes.data) == 0
        H.x264_encoder_close(h_)
        H.x264_picture_clean(C.byref(pic))
        return stream, meta, first, rec

    serial, m1, first1, rec1 = run(0)
    for k in (4, 2):
        par, mk, firstk, reck = run(k)
        assert len(m1) == len(mk) == nfr
        assert mk == m1 and par == serial, "pictures in flight (%d): the stream differs from the serial one" % k
        assert first1 <= firstk <= first1 + k - 1 or firstk == nfr          # at most k - 1 more calls of delay
        assert np.array_equal(reck, rec1)
    dec = O.h264_decode(serial, nfr, w, h)
    assert len(dec) == nfr and np.array_equal(dec[-1], rec1)                # (the last picture in coding order is the one handed back last)

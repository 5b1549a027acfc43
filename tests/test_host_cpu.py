"""CPU: the host side of the drop-in (boundary B1) without a GPU — x264_param_* semantics as the driver uses
them (codec.c:1463,1349,1581,1584), parameter sets, and the CAVLC entropy coder closed-loop against the
checker decoder, fed with macroblock records from the oracle encoder."""
import ctypes as C

import numpy as np
import pytest

import host_lib as HL
import oracle_lib as O
from synth import synth_frames

H = HL.H


def preset(name=None, tune=None):
    p = HL.Param()
    rc = H.x264_param_default_preset(C.byref(p), name.encode() if name else None, tune.encode() if tune else None)
    return rc, p


def test_defaults_are_medium():
    """values the help text prints from x264_param_default (config.c:1544-1692)"""
    rc, p = preset()
    assert rc == 0
    assert (p.i_keyint_max, p.i_scenecut_threshold, p.i_bframe, p.i_bframe_adaptive, p.i_frame_reference) == (250, 40, 3, 1, 3)
    assert (p.b_cabac, p.b_deblocking_filter, p.analyse.i_me_method, p.analyse.i_me_range, p.analyse.i_subpel_refine) == (1, 1, 1, 16, 7)
    assert (p.analyse.i_trellis, p.analyse.b_transform_8x8, p.analyse.i_weighted_pred, p.rc.i_lookahead) == (1, 1, 2, 40)
    assert abs(p.rc.f_rf_constant - 23.0) < 1e-6 and p.rc.i_rc_method == HL.X264_RC_CRF and p.rc.b_mb_tree == 1
    assert (p.analyse.i_luma_deadzone[0], p.analyse.i_luma_deadzone[1]) == (21, 11)


def test_presets_follow_the_reference_table():
    """config.c:1460-1498"""
    _, u = preset("ultrafast")
    assert (u.analyse.b_transform_8x8, u.rc.i_aq_mode, u.i_bframe, u.b_cabac, u.b_deblocking_filter, u.rc.b_mb_tree) == (0, 0, 0, 0, 0, 0)
    assert (u.analyse.i_me_method, u.analyse.inter, u.i_frame_reference, u.analyse.i_subpel_refine, u.analyse.i_weighted_pred) == (0, 0, 1, 0, 0)
    _, s = preset("slow")
    assert (s.analyse.i_direct_mv_pred, s.rc.i_lookahead, s.i_frame_reference, s.analyse.i_subpel_refine, s.analyse.i_trellis) == (3, 50, 5, 8, 2)
    _, v = preset("veryslow")
    assert (v.i_bframe, v.analyse.i_me_method, v.analyse.i_me_range, v.i_frame_reference, v.analyse.i_subpel_refine) == (8, 2, 24, 16, 10)
    assert preset("5")[0] == 0 and preset("nosuch")[0] < 0                 # numeric presets; unknown name -> <0 (codec.c:1463)
    _, z = preset("medium", "zerolatency")
    assert (z.i_bframe, z.rc.i_lookahead, z.rc.b_mb_tree) == (0, 0, 0)
    _, f = preset(None, "fastdecode,zerolatency")                          # comma separated (codec.c:1433-1445)
    assert (f.b_cabac, f.b_deblocking_filter, f.i_bframe) == (0, 0, 0)
    assert preset(None, "film,grain")[0] < 0                               # one psy tuning at a time
    assert preset(None, "nonsense")[0] < 0


def test_param_parse_names_values_and_errors():
    """return codes relied on at codec.c:1357-1361"""
    _, p = preset()
    P = lambda n, v=None: H.x264_param_parse(C.byref(p), n.encode(), v.encode() if v is not None else None)
    assert P("ref", "5") == 0 and p.i_frame_reference == 5
    assert P("no-cabac") == 0 and p.b_cabac == 0
    assert P("cabac") == 0 and p.b_cabac == 1
    assert P("qp", "30") == 0 and p.rc.i_rc_method == HL.X264_RC_CQP and p.rc.i_qp_constant == 30
    assert P("crf", "18.5") == 0 and abs(p.rc.f_rf_constant - 18.5) < 1e-6
    assert P("deblock", "-1:2") == 0 and (p.i_deblocking_filter_alphac0, p.i_deblocking_filter_beta) == (-1, 2)
    assert P("no-deblock") == 0 and p.b_deblocking_filter == 0
    assert P("me", "umh") == 0 and p.analyse.i_me_method == 2
    assert P("partitions", "p8x8,i4x4") == 0 and p.analyse.inter == (0x10 | 0x1)
    assert P("keyint", "infinite") == 0 and p.i_keyint_max == 1 << 30
    assert P("fps", "30000/1001") == 0 and (p.i_fps_num, p.i_fps_den) == (30000, 1001)
    assert P("sar", "4:3") == 0 and (p.vui.i_sar_width, p.vui.i_sar_height) == (4, 3)
    assert P("level", "4.1") == 0 and p.i_level_idc == 41
    assert P("psy-rd", "0.5:0.2") == 0 and abs(p.analyse.f_psy_rd - 0.5) < 1e-6
    assert P("threads", "1") == 0 and P("cpu-independent") == 0 and P("stitchable") == 0
    assert P("this-is-not-an-option", "1") == HL.X264_PARAM_BAD_NAME
    assert P("ref", "abc") == HL.X264_PARAM_BAD_VALUE
    assert P("me", "warp") == HL.X264_PARAM_BAD_VALUE
    assert P("ref") == HL.X264_PARAM_BAD_VALUE                              # missing argument


def test_profile_and_fastfirstpass():
    _, p = preset()
    assert H.x264_param_apply_profile(C.byref(p), None) == 0 and p.b_cabac == 1          # NULL = no restriction (codec.c:1584)
    assert H.x264_param_apply_profile(C.byref(p), b"baseline") == 0
    assert (p.b_cabac, p.i_bframe, p.analyse.b_transform_8x8, p.analyse.i_weighted_pred) == (0, 0, 0, 0)
    assert H.x264_param_apply_profile(C.byref(p), b"nope") < 0
    _, q = preset()
    q.rc.b_stat_write = 1
    H.x264_param_apply_fastfirstpass(C.byref(q))                                           # config.c:1535-1538
    assert (q.i_frame_reference, q.analyse.b_transform_8x8, q.analyse.inter, q.analyse.i_me_method, q.analyse.i_subpel_refine, q.analyse.i_trellis) == (1, 0, 0, 0, 2, 0)


def test_levels_table_and_picture_alloc():
    idcs = [l.level_idc for l in HL.LEVELS]
    assert idcs[-1] == 0 and 40 in idcs and 62 in idcs                      # 0-terminated, up to 6.2 (codec.c:87-89,1596)
    l40 = [l for l in HL.LEVELS if l.level_idc == 40][0]
    assert l40.dpb == 32768 and l40.frame_size == 8192                      # dpb in macroblocks, compared with mbs*refs
    pic = HL.Picture()
    H.x264_picture_clean(C.byref(pic))                                       # safe on a zeroed struct (codec.c:1872)
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, 64, 48) == 0
    assert pic.img.i_plane == 3 and list(pic.img.i_stride)[:3] == [64, 32, 32]
    H.x264_picture_clean(C.byref(pic))
    assert H.x264_picture_alloc(C.byref(pic), 0x7f, 64, 48) < 0


def test_open_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _, p = preset()
    p.i_width, p.i_height = 64, 48
    assert not H.x264_encoder_open_157(C.byref(p))


@pytest.mark.parametrize("w,h,kw", [(64, 48, {}), (176, 144, {}), (176, 144, dict(qp_i=36, qp_p=40)), (208, 120, dict(deblock=0)),
                                     (96, 80, dict(qp_i=8, qp_p=10)), (64, 64, dict(partitions=0)),
                                     (176, 144, dict(partitions=3)), (352, 288, dict(partitions=3, qp_i=30, qp_p=33)),
                                     (208, 120, dict(partitions=1, subme=4)), (176, 144, dict(refs=3, partitions=3)),
                                     (96, 80, dict(refs=2)), (208, 120, dict(refs=4, partitions=3, qp_i=30, qp_p=32)), (96, 80, dict(refs=5, partitions=7, dct8x8=1, mixed_refs=1, qp_i=33, qp_p=36)),
                                     (176, 144, dict(rd=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, subme=7, refs=3, partitions=7, dct8x8=1, mixed_refs=1, chroma_me=1)),
                                     (96, 80, dict(rd=1, subme=6, refs=2, partitions=3, qp_i=30, qp_p=34)),
                                     (176, 144, dict(dct8x8=1)), (352, 288, dict(dct8x8=1, partitions=3, refs=2, qp_i=26, qp_p=28)),
                                     (208, 120, dict(dct8x8=1, qp_i=12, qp_p=14, dct_decimate=0)),
                                     (176, 144, dict(dct8x8=1, partitions=6)), (352, 288, dict(dct8x8=1, partitions=7, refs=3, qp_i=28, qp_p=31)),
                                     (208, 120, dict(dct8x8=1, partitions=4, qp_i=14, qp_p=16)), (64, 48, dict(dct8x8=1, partitions=7, qp_i=38, qp_p=40))])
@pytest.mark.parametrize("cabac", [0, 1])
def test_entropy_closed_loop(w, h, kw, cabac):
    """oracle records -> host CAVLC / CABAC -> checker decoder == oracle reconstruction, I and P pictures"""
    nfr = 7 if kw.get("refs", 1) > 1 else 4
    frames = synth_frames(w, h, nfr, seed=11 * w + h)
    cfg = O.default_config(w, h, **kw)
    enc = O.OracleEncoder(cfg)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, cqo=cfg.chroma_qp_offset, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)
    recons, skipped = [], 0
    for i, f in enumerate(frames):
        idr = i == 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        s, sk = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0,
                               0 if cfg.deblock else 1, mbs, lv, num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)
        if cfg.refs > 1 and i >= 2:
            assert (mbs["ref"][mbs["type"] >= 4] >= 0).all()
        stream += s
        skipped += sk
        recons.append(enc.recon())
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"picture {i}")


@pytest.mark.parametrize("cabac", [0, 1])
@pytest.mark.parametrize("w,h,slices,kw", [(176, 144, 2, dict(partitions=3, refs=2)), (96, 208, 3, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5)),
                                          (64, 272, 4, dict(aq_mode=1, partitions=7, dct8x8=1, qp_i=30, qp_p=34)), (208, 128, 2, dict(deblock=0, subme=2)),
                                          (48, 336, 5, dict(partitions=3, refs=2, qp_i=12, qp_p=15, dct_decimate=0))])
def test_sliced_pictures_closed_loop(w, h, slices, kw, cabac):
    """x264's slice threads: N slices per picture, each analysed on its own (nothing above the slice's first row is available, own fast-intra
    statistics and quantiser chain) and written as its own NAL with disable_deblocking_filter_idc 2: oracle records -> host writers ->
    checker decoder (slice-membership availability, no filtering across slice edges) == oracle reconstruction"""
    nfr = 5
    frames = synth_frames(w, h, nfr, seed=3 * w + h)
    cfg = O.default_config(w, h, slices=slices, **kw)
    enc, one = O.OracleEncoder(cfg), O.OracleEncoder(O.default_config(w, h, **kw))
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)
    recons, differs = [], False
    for i, f in enumerate(frames):
        idr = i == 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        m1, _ = one.encode(f, 2 if idr else 0)
        differs |= not np.array_equal(mbs.view(np.uint8), m1.view(np.uint8))
        s, _ = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0,
                              0 if cfg.deblock else 1, mbs, lv, num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac, slices=slices)
        assert s.count(b"\x00\x00\x01") >= slices                 # one NAL per slice
        stream += s
        recons.append(enc.recon())
    assert differs                                                    # slicing changes decisions (it must: predictions stop at slice edges)
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"picture {i}")


@pytest.mark.parametrize("cabac", [0, 1])
@pytest.mark.parametrize("w,h,slices,kw", [(176, 144, 9, dict(partitions=3, refs=2)), (96, 208, 5, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5)),
                                          (64, 272, 17, dict(aq_mode=1, partitions=7, dct8x8=1, qp_i=30, qp_p=34)), (208, 128, 3, dict(qp_i=34, qp_p=38, subme=2))])
def test_plain_slices_closed_loop(w, h, slices, kw, cabac):
    """x264's --slices N (slices_plain): the split of slice threads, down to one macroblock row per slice, but the loop filter crosses the slice
    boundaries (disable_deblocking_filter_idc 0) and the intra statistics run on through the picture: oracle records -> host writers ->
    checker decoder == oracle reconstruction; an I picture differs from the slice-thread one exactly by that filtering"""
    nfr = 4
    frames = synth_frames(w, h, nfr, seed=5 * w + h)
    cfg = O.default_config(w, h, slices=slices, slices_plain=1, **kw)
    enc = O.OracleEncoder(cfg)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    thr = O.OracleEncoder(O.default_config(w, h, slices=slices, **kw)) if slices <= mbh // 4 else None
    stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)
    recons = []
    for i, f in enumerate(frames):
        idr = i == 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        s, _ = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0,
                              0, mbs, lv, num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac, slices=-slices)
        assert s.count(b"\x00\x00\x01") >= slices
        stream += s
        recons.append(enc.recon())
        if thr is not None and idr:
            m2, _ = thr.encode(f, 2)
            np.testing.assert_array_equal(mbs.view(np.uint8), m2.view(np.uint8))        # the first picture's decisions do not see the filter ...
            assert not np.array_equal(thr.recon(), recons[0])                           # ... its reconstruction does
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    for i in range(nfr):
        np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"picture {i}")


def test_cavlc_static_sequence_uses_skip():
    w, h = 96, 64
    f = synth_frames(w, h, 1, seed=5)[0]
    cfg = O.default_config(w, h, qp_i=20, qp_p=30)
    enc = O.OracleEncoder(cfg)
    mbs, lv = enc.encode(f, 2)
    stream = HL.write_headers(w, h, pic_init_qp=30) + HL.write_slice(6, 4, 2, 20, 30, 0, 8, 1, 0, 0, mbs, lv)[0]
    rec = enc.recon()
    mbs, lv = enc.encode(rec, 0)
    s, sk = HL.write_slice(6, 4, 0, 30, 30, 1, 8, 0, 0, 0, mbs, lv)
    assert sk == 24 and len(s) < 16                                          # every macroblock is P_Skip
    dec = O.h264_decode(stream + s, 2, w, h)
    np.testing.assert_array_equal(dec[1], enc.recon())


def test_cavlc_tables_are_prefix_codes():
    """structural check of Tables 9-5/9-7/9-8/9-9/9-10 as typed in cavlc_tables.hpp"""
    import re
    from fractions import Fraction
    import os
    src = open(os.path.join(HL.ROOT, "x264vfw_amd", "host", "cavlc_tables.hpp")).read()

    def arr(name):
        body = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", src, re.S).group(1)
        rows = re.findall(r"\{([^{}]*)\}", body) or [body]
        return [[int(x) for x in r.replace("\n", " ").split(",") if x.strip()] for r in rows]

    def check(lens, bits, complete):
        codes = [format(b, "0%db" % l) for l, b in zip(lens, bits) if l > 0]
        assert len(set(codes)) == len(codes)
        assert not any(a != b and b.startswith(a) for a in codes for b in codes)
        k = sum(Fraction(1, 2 ** len(c)) for c in codes)
        assert k == 1 if complete else k < 1
    for t in range(4):
        check(arr("coeff_token_len")[t], arr("coeff_token_bits")[t], False)
    check(arr("chroma_dc_coeff_token_len")[0], arr("chroma_dc_coeff_token_bits")[0], True)
    for t in range(15):
        check(arr("total_zeros_len")[t], arr("total_zeros_bits")[t], t > 0)
    for t in range(3):
        check(arr("chroma_dc_total_zeros_len")[t][:4 - t], arr("chroma_dc_total_zeros_bits")[t][:4 - t], True)
    for t in range(7):
        check(arr("run_before_len")[t], arr("run_before_bits")[t], t < 6)
    assert sorted(arr("cbp_to_golomb_intra")[0]) == list(range(48)) == sorted(arr("cbp_to_golomb_inter")[0])


def test_host_abi_exports_every_declared_symbol():
    """include/x264.h (B1), include/vfw_shim.h (B2) and include/x264gpu_host.h are contracts: every function they declare is
    exported by libx264gpu_host.so (no compute calls here — there is no GPU on this box)"""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    declared = set()
    for hdr, pat in (("x264.h", r"\b(x264_[a-z0-9_]+)\s*\("), ("vfw_shim.h", r"\b(DriverProc|x264vfw_[a-z0-9_]+)\s*\("),
                     ("x264gpu_host.h", r"\b(x264gpu_host_[a-z0-9_]+|x264host_[a-z0-9_]+)\s*\(")):
        text = open(os.path.join(root, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(pat, text))
    declared.discard("x264_encoder_open")                       # a macro onto x264_encoder_open_<build> (codec.c:1623)
    assert len(declared) > 15, declared
    missing = [d for d in sorted(declared) if not hasattr(H, d)]
    assert not missing, f"declared but not exported: {missing}"


@pytest.mark.parametrize("cabac", [0, 1])
@pytest.mark.parametrize("seed", [21, 22, 23])
def test_entropy_closed_loop_random(seed, cabac):
    """randomised closed loop on the CPU: oracle records -> host CAVLC / CABAC -> checker decoder == oracle reconstruction, over random
    sizes, quantisers 0..51 and toolset combinations (partitions, refs, 8x8 transform, Intra_8x8, dia/hex, deblock offsets)"""
    import random
    rnd = random.Random(seed)
    for it in range(12):
        w = max(16, 16 * rnd.randint(1, 9) - rnd.choice([0, 0, 2, 6, 14]))
        h = max(16, 16 * rnd.randint(1, 7) - rnd.choice([0, 0, 2, 8, 12]))
        dct = rnd.randint(0, 1)
        kw = dict(refs=rnd.randint(1, 5), partitions=rnd.choice([0, 1, 2, 3, 4, 5, 6, 7]) if dct else rnd.choice([0, 1, 2, 3]), dct8x8=dct,
                  subme=rnd.choice([0, 2, 5, 7]), me_method=rnd.randint(0, 3), chroma_me=rnd.randint(0, 1), mixed_refs=rnd.randint(0, 1), aq_mode=rnd.randint(0, 1), aq_strength=rnd.choice([0.51985, 1.0397, 1.55955]), qp_i=rnd.randint(0, 51), qp_p=rnd.randint(0, 51),
                  deblock=rnd.randint(0, 1), dct_decimate=rnd.randint(0, 1), chroma_qp_offset=rnd.randint(-6, 6))
        nfr = rnd.randint(2, 6)
        frames = synth_frames(w, h, nfr, seed=rnd.randint(0, 10 ** 6))
        cfg = O.default_config(w, h, **kw)
        enc = O.OracleEncoder(cfg)
        mbw, mbh = (w + 15) // 16, (h + 15) // 16
        stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, cqo=cfg.chroma_qp_offset, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)
        recons = []
        for i, f in enumerate(frames):
            idr = i == 0
            mbs, lv = enc.encode(f, 2 if idr else 0)
            stream += HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0,
                                     0 if cfg.deblock else 1, mbs, lv, num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=cabac)[0]
            recons.append(enc.recon())
        dec = O.h264_decode(stream, nfr, w, h)
        assert len(dec) == nfr, f"seed {seed} case {it}: {w}x{h} {kw}: decoder returned {len(dec)} pictures"
        for i in range(nfr):
            np.testing.assert_array_equal(dec[i], recons[i], err_msg=f"seed {seed} case {it}: {w}x{h} {kw} picture {i}")


def test_row_band_parallel_cavlc_is_byte_identical(monkeypatch):
    """write_slice codes row bands on several threads and stitches the bit strings (mb_skip_run carried across bands): the NAL
    must not depend on the thread count — I and P slices, skip runs crossing band borders, ragged row counts."""
    import random
    rnd = random.Random(3)
    for w, h, kw in [(352, 288, dict(qp_i=30, qp_p=38)), (176, 144, dict(qp_i=20, qp_p=23, partitions=7, dct8x8=1, refs=2)),
                     (208, 120, dict(qp_i=45, qp_p=50)), (320, 368, dict(qp_i=26, qp_p=51, partitions=3))]:
        mbw, mbh = (w + 15) // 16, (h + 15) // 16
        enc = O.OracleEncoder(O.default_config(w, h, **kw))
        frames = synth_frames(w, h, 3, seed=rnd.randint(0, 999))
        frames.append(frames[-1].copy())                       # a repeated picture: long skip runs
        for i, f in enumerate(frames):
            st = 2 if i == 0 else 0
            mbs, lv = enc.encode(f, st)
            outs = []
            for t in ("1", "2", "3", "7"):
                monkeypatch.setenv("X264GPU_CAVLC_THREADS", t)
                nal, skipped = HL.write_slice(mbw, mbh, st, kw["qp_i"] if st == 2 else kw["qp_p"], 26, i, 4, int(st == 2), 0, 0, mbs, lv,
                                              num_ref=max(1, min(i, kw.get("refs", 1))), num_ref_default=kw.get("refs", 1), t8x8=kw.get("dct8x8", 0))
                outs.append((nal, skipped))
            assert all(o == outs[0] for o in outs), f"{w}x{h} frame {i}"
        assert outs[0][1] > mbw                                # the repeated picture really is mostly skipped


def test_cabac_context_tables_typed_twice_agree():
    """the (m, n) context initialisation values exist twice, typed separately: per context index in the product's encoder
    (x264vfw_amd/host/cabac_tables.hpp) and per syntax element in the checker decoder (oracle/cabac_dec.hpp); every context a 4:2:0
    I / P stream uses must carry the same pair in both (typing errors; neither copy could be checked against the standard's text here)"""
    import ctypes as C
    O.L.x264o_cabac_tables_mismatches.restype = C.c_int
    assert O.L.x264o_cabac_tables_mismatches() == 0


def test_cavlc_tables_typed_twice_agree():
    """the CAVLC code tables exist twice, typed separately: (length, value) arrays in the product's writer (x264vfw_amd/host/cavlc_tables.hpp)
    and bit strings per symbol, as the standard prints them, in the checker decoder (oracle/cavlc_dec.hpp).  Every symbol must carry the same
    code in both, and each of the decoder's tables must be a prefix code"""
    import ctypes as C
    for f in (O.L.x264o_cavlc_tables_mismatches, O.L.x264o_cavlc_tables_prefix_clashes):
        f.restype = C.c_int
        assert f() == 0


@pytest.mark.parametrize("w,h,kw", [(176, 144, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5)), (96, 80, dict(partitions=3, refs=2, qp_i=14, qp_p=16)),
                                    (208, 120, dict(aq_mode=1, partitions=7, dct8x8=1, refs=2, qp_i=30, qp_p=33)), (64, 208, dict(slices=3, partitions=7, dct8x8=1, refs=2))])
def test_rd_bit_counts_equal_the_bits_the_decoder_consumes(w, h, kw):
    """the CAVLC bit count behind the RD costs (oracle mb_bits_cavlc: x264_macroblock_size_cavlc) of every final macroblock equals what the
    checker decoder consumes for that macroblock's layer in the stream the host writer produced — mb_type, references, vector differences
    (true predictors), intra modes, cbp, transform flag, mb_qp_delta (AQ case), every residual block with its nC"""
    import ctypes as C
    nfr = 4
    frames = synth_frames(w, h, nfr, seed=w + 7 * h)
    cfg = O.default_config(w, h, **kw)
    enc = O.OracleEncoder(cfg)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    pred = np.zeros(mbw * mbh, np.int32)
    O.L.x264o_encoder_set_mb_bits_out.argtypes = [C.c_void_p, C.c_void_p]
    O.L.x264o_encoder_set_mb_bits_out(enc.h, pred.ctypes.data)
    stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=0)
    want = []
    for i, f in enumerate(frames):
        idr = i == 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        want.append(pred.copy())
        stream += HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0, 0, mbs, lv,
                                 num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=0, slices=kw.get("slices", 1))[0]
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    got = np.zeros(nfr * mbw * mbh, np.int32)
    O.L.x264o_h264_last_mb_bits.restype = C.c_int
    assert O.L.x264o_h264_last_mb_bits(got.ctypes.data_as(C.c_void_p), got.size) == got.size
    np.testing.assert_array_equal(got.reshape(nfr, -1), np.stack(want))


@pytest.mark.parametrize("w,h,kw", [(176, 144, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
                                     (96, 80, dict(partitions=3, refs=2, qp_i=14, qp_p=16)), (96, 208, dict(slices=1, partitions=7, dct8x8=1, aq_mode=1, qp_i=30, qp_p=33)),
                                     (64, 48, dict(partitions=0, qp_i=40, qp_p=44))])
def test_cabac_rd_states_and_sizes_follow_the_real_coder(w, h, kw):
    """CABAC RD (x264 subme 6 / 7 with cabac): (1) the context states the oracle's macroblock loop carries (cabac_rd.cpp "evolve": the
    finished macroblocks' bins, states only) equal, after every picture, the states the product's arithmetic coder ends the slice with — the
    states a decoder holds, which the checker decoder proves by decoding that slice; (2) the size estimates (entropy table, 1/256 bit)
    summed over a picture stay within a few percent of the bytes the arithmetic coder really wrote"""
    import ctypes as C
    nfr = 4
    frames = synth_frames(w, h, nfr, seed=3 * w + h)
    cfg = O.default_config(w, h, cabac=1, rd=1, subme=7, **kw)
    enc = O.OracleEncoder(cfg)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    est = np.zeros(mbw * mbh, np.int32)
    O.L.x264o_encoder_set_mb_bits_out.argtypes = [C.c_void_p, C.c_void_p]
    O.L.x264o_encoder_set_mb_bits_out(enc.h, est.ctypes.data)
    O.L.x264o_encoder_cabac_states.argtypes = [C.c_void_p, C.c_void_p]
    stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=1, cqo=cfg.chroma_qp_offset)
    total_est = total_real = 0
    for i, f in enumerate(frames):
        idr = i == 0
        est[:] = 0
        mbs, lv = enc.encode(f, 2 if idr else 0)
        nal = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0, 0, mbs, lv,
                             num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=1)[0]
        stream += nal
        got, want = np.zeros(460, np.uint8), np.zeros(460, np.uint8)
        O.L.x264o_encoder_cabac_states(enc.h, got.ctypes.data)
        HL.H.x264host_cabac_last_states(want.ctypes.data_as(C.c_void_p))
        # the contexts the slice type uses (the product seeds every table row, the checker only these)
        used = list(range(3, 11)) + list(range(60, 70)) + list(range(73, 276)) + list(range(399, 436)) + ([] if idr else list(range(11, 24)) + list(range(40, 60)))
        np.testing.assert_array_equal(got[used], want[used], err_msg=f"context states after picture {i}")
        total_est += int(est.sum()) / 256.0
        total_real += 8 * len(nal)
    assert len(O.h264_decode(stream, nfr, w, h)) == nfr
    assert abs(total_est - total_real) < 0.06 * total_real + 64 * nfr, (total_est, total_real)


@pytest.mark.parametrize("w,h,kw", [(176, 144, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, chroma_me=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
                                     (176, 144, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, chroma_me=1, rd=1, subme=7)),
                                     (176, 144, dict(partitions=3, refs=1, subme=5)), (96, 80, dict(partitions=7, dct8x8=1, qp_i=12, qp_p=14, subme=5)),
                                     (96, 80, dict(partitions=7, dct8x8=1, qp_i=40, qp_p=43, rd=1, subme=6))])
def test_oracle_trellis_streams_decode_and_pay_off(w, h, kw):
    """trellis quantisation in the checker (oracle/trellis.cpp, x264 --trellis 1; the device does not have it yet): the levels it picks are
    ordinary levels — the stream written from them decodes to the encoder's own reconstruction — and the search pays off in rate-distortion
    terms: PSNR gained plus the bits saved (at this content's ~3.3 dB per doubling of the rate) is positive.  With psy-RD on the mode
    decision pulls the other way (it buys back the energy the trellis removed), so that case only checks the stream"""
    nfr = 6
    frames = synth_frames(w, h, nfr, seed=5 * w + h)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    res = {}
    for trellis in (0, 1):
        cfg = O.default_config(w, h, cabac=1, trellis=63 * trellis, **kw)
        enc = O.OracleEncoder(cfg)
        stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=1, cqo=cfg.chroma_qp_offset)
        recs, bits = [], 0
        for i, f in enumerate(frames):
            idr = i == 0
            mbs, lv = enc.encode(f, 2 if idr else 0)
            nal = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0, 0, mbs, lv,
                                 num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=1)[0]
            stream += nal; bits += 8 * len(nal)
            recs.append(enc.recon().copy())
        dec = O.h264_decode(stream, nfr, w, h)
        assert len(dec) == nfr
        sse = 0.0
        for i in range(nfr):
            np.testing.assert_array_equal(dec[i], recs[i], err_msg=f"trellis {trellis} picture {i}")
            sse += float(((recs[i][:w * h].astype(np.int64) - frames[i][:w * h].astype(np.int64)) ** 2).sum())
        res[trellis] = (bits, 10 * np.log10(255.0 ** 2 * w * h * nfr / max(sse, 1.0)))
    (b0, p0), (b1, p1) = res[0], res[1]
    assert b1 != b0, res
    if not kw.get("psy"):
        assert (p1 - p0) + 3.3 * np.log2(b0 / b1) > 0.05, res


@pytest.mark.parametrize("w,h,kw", [(176, 144, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, chroma_me=1, rd=1)),
                                     (176, 144, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, rd=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63)),
                                     (96, 80, dict(partitions=7, dct8x8=0, qp_i=34, qp_p=37, rd=1)),
                                     (208, 120, dict(partitions=7, dct8x8=1, qp_i=14, qp_p=16, rd=1, refs=2, trellis=127))])
def test_oracle_rd_refinement_streams_decode_and_pay_off(w, h, kw):
    """RD refinement in the checker (x264 --subme 8: x264_me_refine_qpel_rd of the chosen P partitions, intra_rd_refine of the chosen intra type's
    modes; oracle/analyse.c): the refined vectors and modes are ordinary ones — the stream written from them decodes to the encoder's own
    reconstruction — and without psy-RD (which trades PSNR for energy on purpose) the refinement wins in rate-distortion terms over subme 7"""
    nfr = 6
    frames = synth_frames(w, h, nfr, seed=5 * w + h)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    res = {}
    for sub in (7, 8):
        cfg = O.default_config(w, h, cabac=1, subme=sub, **dict(kw, rd=63 if sub == 8 else 1))
        enc = O.OracleEncoder(cfg)
        stream = HL.write_headers(w, h, pic_init_qp=cfg.qp_p, num_ref=cfg.refs, t8x8=cfg.dct8x8, cabac=1, cqo=cfg.chroma_qp_offset)
        recs, bits = [], 0
        for i, f in enumerate(frames):
            idr = i == 0
            mbs, lv = enc.encode(f, 2 if idr else 0)
            nal = HL.write_slice(mbw, mbh, 2 if idr else 0, cfg.qp_i if idr else cfg.qp_p, cfg.qp_p, i, 8, int(idr), 0, 0, mbs, lv,
                                 num_ref=max(1, min(cfg.refs, i)), num_ref_default=cfg.refs, t8x8=cfg.dct8x8, cabac=1)[0]
            stream += nal; bits += 8 * len(nal)
            recs.append(enc.recon().copy())
        dec = O.h264_decode(stream, nfr, w, h)
        assert len(dec) == nfr
        sse = 0.0
        for i in range(nfr):
            np.testing.assert_array_equal(dec[i], recs[i], err_msg=f"subme {sub} picture {i}")
            sse += float(((recs[i][:w * h].astype(np.int64) - frames[i][:w * h].astype(np.int64)) ** 2).sum())
        res[sub] = (bits, 10 * np.log10(255.0 ** 2 * w * h * nfr / max(sse, 1.0)))
    (b0, p0), (b1, p1) = res[7], res[8]
    assert b1 != b0, res
    if not kw.get("psy"):
        assert (p1 - p0) + 3.3 * np.log2(b0 / b1) > 0.0, res

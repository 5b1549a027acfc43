#!/usr/bin/env python3
"""A session with B pictures through x264_encoder_encode() of the STUB-backed host library: prints one JSON line with the picture types, pts / dts of
the pictures as they leave, and writes the stream to the given file.  Usage: run_host_b.py OUT W H FRAMES SEED key=value ..."""
import ctypes as C
import json
import os
import sys

os.environ["X264_HOST_STUB"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import host_lib as HL  # noqa: E402
from synth import synth_frames  # noqa: E402


def make_frames(w, h, n, seed, scene_len=0, static=0, fade=0):
    """the clips of the session tests: synthetic frames, optionally with scene cuts every scene_len frames, fading to black by `fade` per cent a
    picture (tests/test_bframes_cpu.py fade_frames), or the first picture again and again with a little noise"""
    import numpy as np
    frames = synth_frames(w, h, n, seed=seed, **({"scene_len": scene_len} if scene_len else {}))
    if fade:
        for i, f in enumerate(frames):
            a = max(0.0, 1.0 - i * fade / 100.0)
            g = f.astype(np.float32)
            g[:w * h] *= a
            g[w * h:] = 128 + (g[w * h:] - 128) * a
            frames[i] = np.clip(np.rint(g), 0, 255).astype(np.uint8)
    if static:
        rng = np.random.default_rng(seed)
        frames = [np.clip(frames[0].astype(np.int16) + rng.integers(-1, 2, frames[0].shape), 0, 255).astype(np.uint8) for _ in range(n)]
    return frames


def main():
    out_path = sys.argv[1]
    w, h, n, seed = (int(x) for x in sys.argv[2:6])
    opts = {}
    for a in sys.argv[6:]:
        k, _, v = a.partition("=")
        opts[k] = v if _ else None
    H = HL.H
    scene_len = int(opts.pop("scene_len", 0) or 0)
    static = int(opts.pop("static", 0) or 0)
    fade = int(opts.pop("fade", 0) or 0)
    preset = (opts.pop("preset", None) or "medium").encode()
    want_log = int(opts.pop("log", 0) or 0)
    want_plan = int(opts.pop("plan", 0) or 0)          # a second pass: also print the quantiser scales init_pass2 planned (x264host_pass2_plan)
    frames = make_frames(w, h, n, seed, scene_len, static, fade)
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), preset, None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, (k, v)
    log = []
    cb = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p, C.c_void_p)(lambda priv, lvl, fmt, va: log.append([lvl, fmt.decode(errors="replace").strip()]))
    if want_log:          # the format strings of what the session reports through pf_log (codec.c:1274-1283 routes them to the driver's log window)
        p.pf_log = C.cast(cb, C.c_void_p).value
        p.i_log_level = 3
    p.b_vfr_input = 0                                                  # the driver forces constant frame rate (codec.c:1476-1480)
    p.b_annexb, p.b_repeat_headers = 1, 1
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    eff = HL.Param()
    H.x264_encoder_parameters(h_, C.byref(eff))
    plan = None
    if want_plan:
        qs, eb = (C.c_double * n)(), (C.c_double * n)()
        cnt = H.x264host_pass2_plan(h_, qs, eb, n)
        plan = {"count": cnt, "new_qscale": list(qs)[:cnt], "expected_bits": list(eb)[:cnt]}
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, nn = C.POINTER(HL.Nal)(), C.c_int()
    planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
    stream, recs = b"", []
    first_out = None

    def take(size):
        nonlocal stream
        if size > 0:
            stream += C.string_at(nal[0].p_payload, size)
            recs.append((out.i_type, out.i_pts, out.i_dts, out.b_keyframe, size))
    for i, f in enumerate(frames):
        for pl, (sz, off) in enumerate(planes):
            C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
        pic.i_pts = i
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
        assert size >= 0
        if size > 0 and first_out is None:
            first_out = i + 1
        take(size)
    while H.x264_encoder_delayed_frames(h_):
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
        assert size > 0
        take(size)
    H.x264_encoder_close(h_)
    open(out_path, "wb").write(stream)
    print(json.dumps({"recs": recs, "bframes": eff.i_bframe, "pyramid": eff.i_bframe_pyramid, "badapt": eff.i_bframe_adaptive, "weightb": eff.analyse.b_weighted_bipred,
                      "weightp": eff.analyse.i_weighted_pred, "mbtree": eff.rc.b_mb_tree, "subme": eff.analyse.i_subpel_refine, "cabac": eff.b_cabac, "direct": eff.analyse.i_direct_mv_pred, "first_output_after": first_out, "log": log,
                      "rc_method": eff.rc.i_rc_method, "stat_read": eff.rc.b_stat_read, "stat_write": eff.rc.b_stat_write, "inter": eff.analyse.inter, "refs": eff.i_frame_reference,
                      "me": eff.analyse.i_me_method, "trellis": eff.analyse.i_trellis, "mv_range": eff.analyse.i_mv_range, "me_range": eff.analyse.i_me_range, "plan": plan}))


if __name__ == "__main__":
    main()

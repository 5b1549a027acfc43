#!/usr/bin/env python3
"""PCIe-inclusive rate of the B1 boundary: ONE stream through x264_encoder_encode() with host pictures (upload, GPU hot
path, record/level download, host CAVLC), the way the VfW driver calls it (codec.c:1693).  Not the bench.py metric — that
one keeps inputs resident in HBM and many streams in flight; this is the single-stream latency-bound figure DESIGN.md quotes."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from x264vfw_amd import host_api as HL  # noqa: E402
from x264vfw_amd.synth import synth_frames  # noqa: E402

H = HL.H


def main():
    w, h, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 60
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 1              # --threads G: closed GOPs coded in lock-step
    keyint = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    rc = sys.argv[4] if len(sys.argv) > 4 else "qp"                     # "qp": constant quantiser; "crf": the driver's default session (CRF 23 + AQ + mbtree, rc-lookahead 40)
    frames = synth_frames(w, h, 8, seed=1)
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
    for k, v in ((("qp", "23") if rc == "qp" else ("crf", "23")), ("keyint", str(keyint)), ("threads", str(threads))):
        assert H.x264_param_parse(C.byref(p), k.encode(), v.encode()) == 0
    for kv in sys.argv[5:]:                                             # further x264 options: slices=68, sliced-threads, ...
        k, _, v = kv.partition("=")
        assert H.x264_param_parse(C.byref(p), k.encode(), v.encode() if v else None) == 0, kv
    p.b_annexb, p.b_repeat_headers = 1, 1
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, nn = C.POINTER(HL.Nal)(), C.c_int()
    planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
    total, t_enc = 0, 0.0
    got = 0
    t_start = time.perf_counter()
    for i in range(n):
        f = frames[i % len(frames)]
        for pl, (sz, off) in enumerate(planes):
            C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
        pic.i_pts = i
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
        assert size >= 0
        got += size > 0
        total += size
    while H.x264_encoder_delayed_frames(h_):
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
        assert size > 0
        got += 1
        total += size
    t_enc = time.perf_counter() - t_start
    H.x264_encoder_close(h_)
    assert got == n
    print(f"B1 single stream 1080p medium toolset, {rc}, threads {threads}, keyint {keyint}: {n / t_enc:.1f} frames/s ({1e3 * t_enc / n:.2f} ms/frame incl. "
          f"host copy-in, upload, GPU, download, host CAVLC; {total / n / 1e3:.1f} kB/frame; {os.cpu_count()} host cores)")


if __name__ == "__main__":
    main()

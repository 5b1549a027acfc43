#!/usr/bin/env python3
"""One 1080p session through x264_encoder_encode with a line per call: seconds inside the call, the type and the pts of the picture handed back — the timeline of a session
with pictures in flight (X264GPU_INFLIGHT=<n>, 0: one picture a call).  Under `rocprofv3 --kernel-trace` the k_mb_slice rows give the device's side of it
(profiles/r06_inflight_kernel_timeline.txt).   Usage: inflight_probe.py FRAMES key=value ...   (x264 options, e.g. qp=27 threads=1 b-adapt=0 scenecut=0)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import host_lib as HL
from synth import synth_frames
H = HL.H
w, h = 1920, 1080
n = int(sys.argv[1]); opts = sys.argv[2:]
src = synth_frames(w, h, 16, seed=0x264, scene_len=97)
p = HL.Param()
assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
for o in opts:
    k, _, v = o.partition("=")
    assert H.x264_param_parse(C.byref(p), k.encode(), v.encode() if v else None) == 0, o
p.b_annexb, p.b_repeat_headers = 1, 1
h_ = H.x264_encoder_open_157(C.byref(p)); assert h_
pic, out = HL.Picture(), HL.Picture()
assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
nal, nn = C.POINTER(HL.Nal)(), C.c_int()
planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
t0 = time.perf_counter(); log = []
for i in range(n):
    f = src[i % len(src)]
    for pl, (sz, off) in enumerate(planes): C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
    pic.i_pts = i
    t = time.perf_counter()
    size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
    log.append((i, round(time.perf_counter() - t, 3), size, int(out.i_type) if size > 0 else -1, int(out.i_pts) if size > 0 else -1))
while H.x264_encoder_delayed_frames(h_):
    t = time.perf_counter()
    size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
    log.append(("f", round(time.perf_counter() - t, 3), size, int(out.i_type), int(out.i_pts)))
dt = time.perf_counter() - t0
H.x264_encoder_close(h_)
print("INFLIGHT", os.environ.get("X264GPU_INFLIGHT"), opts, "fps %.3f" % (n / dt))
print(" ".join("%s:%.2f/t%d/p%d" % (a, b, d, e) for a, b, c, d, e in log))

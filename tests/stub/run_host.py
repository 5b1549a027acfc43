#!/usr/bin/env python3
"""Encodes a synthetic sequence through x264_encoder_encode() of the STUB-backed host library (tests/stub/) and prints one JSON line:
stream hash, sizes, x264gpu_encode_frames calls per stub device.  Run as a child process (the stub shares its soname with the real
device library).  Usage: run_host.py W H FRAMES SEED key=value ...   (x264 options; env X264GPU_STUB_DEVICES = devices the stub shows)"""
import ctypes as C
import hashlib
import json
import os
import sys

os.environ["X264_HOST_STUB"] = "1"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import host_lib as HL  # noqa: E402
from synth import synth_frames  # noqa: E402


def encode(w, h, n, seed, opts, preset=b"medium"):
    H = HL.H
    frames = synth_frames(w, h, n, seed=seed)
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), preset, None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, int(os.environ.get("X264_HOST_LOG", "-1"))
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, (k, v)
    p.b_annexb, p.b_repeat_headers = 1, 1
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, nn = C.POINTER(HL.Nal)(), C.c_int()
    planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
    stream, sizes, meta = b"", [], []

    def take(size):
        nonlocal stream
        if size > 0:
            stream += C.string_at(nal[0].p_payload, size)
            sizes.append(size)
            meta.append((int(out.i_type), int(out.b_keyframe), int(out.i_pts), int(out.i_dts), [(int(nal[k].i_type), int(nal[k].i_ref_idc)) for k in range(nn.value)]))
    for i, f in enumerate(frames):
        for pl, (sz, off) in enumerate(planes):
            C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
        pic.i_pts = i
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(out))
        assert size >= 0
        take(size)
    while H.x264_encoder_delayed_frames(h_):
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(out))
        assert size > 0
        take(size)
    H.x264_encoder_close(h_)
    stub = C.CDLL(os.path.join(HERE, "_build", "libx264gpu.so"))
    stub.x264gpu_stub_encode_calls.restype = C.c_long
    calls = [stub.x264gpu_stub_encode_calls(d) for d in range(int(os.environ.get("X264GPU_STUB_DEVICES", "2")))]
    stub.x264gpu_stub_views.restype = C.c_long
    return {"sha": hashlib.sha256(stream).hexdigest(), "bytes": len(stream), "frames": len(sizes), "calls": calls, "views": stub.x264gpu_stub_views(),
            "meta": hashlib.sha256(json.dumps(meta).encode()).hexdigest(), "pts": [m[2] for m in meta], "dts": [m[3] for m in meta]}, stream


if __name__ == "__main__":
    w, h, n, seed = (int(x) for x in sys.argv[1:5])
    opts = {}
    for a in sys.argv[5:]:
        k, _, v = a.partition("=")
        opts[k] = v if _ else None
    print(json.dumps(encode(w, h, n, seed, opts)[0]))

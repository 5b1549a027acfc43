#!/usr/bin/env python3
"""N sessions in N threads with X264GPU_BATCH=N (the cross-session batcher of the product's host library over the stand-in device), then the same N
sessions one after the other without it: prints one JSON line {"equal": [...], "sizes": [...], "types": "..."}.
Usage: run_host_batch.py N W H FRAMES key=value ...   (X264_HOST_STUB selects the stand-in device; without it the real device is used)"""
import ctypes as C
import json
import os
import sys
import threading

HERE = os.path.dirname(os.path.abspath(__file__))
if os.environ.get("X264_HOST_STUB", "1") == "1" and "--gpu" not in sys.argv:
    os.environ["X264_HOST_STUB"] = "1"
else:
    os.environ.pop("X264_HOST_STUB", None)
sys.argv = [a for a in sys.argv if a != "--gpu"]
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import host_lib as HL  # noqa: E402
from synth import synth_frames  # noqa: E402


def session(w, h, frames, opts, out, idx, errs):
    try:
        H = HL.H
        p = HL.Param()
        assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
        p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
        p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
        for k, v in opts.items():
            assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, (k, v)
        p.b_annexb, p.b_repeat_headers = 1, 1
        h_ = H.x264_encoder_open_157(C.byref(p))
        assert h_
        pic, po = HL.Picture(), HL.Picture()
        assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
        nal, nn = C.POINTER(HL.Nal)(), C.c_int()
        planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
        stream, types = b"", []
        for i, f in enumerate(frames):
            for pl, (sz, off) in enumerate(planes):
                C.memmove(pic.img.plane[pl], f[off:off + sz].ctypes.data, sz)
            pic.i_pts = i
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), C.byref(pic), C.byref(po))
            assert size >= 0, "encode failed"
            if size:
                stream += C.string_at(nal[0].p_payload, size); types.append(po.i_type)
        while H.x264_encoder_delayed_frames(h_):
            size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(nn), None, C.byref(po))
            assert size > 0, "flush failed"
            stream += C.string_at(nal[0].p_payload, size); types.append(po.i_type)
        H.x264_encoder_close(h_)
        out[idx] = (stream, types)
    except Exception as e:  # noqa: BLE001
        errs.append(f"session {idx}: {e!r}")


def main():
    n, w, h, nf = (int(x) for x in sys.argv[1:5])
    opts = {}
    for a in sys.argv[5:]:
        k, _, v = a.partition("=")
        opts[k] = v if _ else None
    clips = [synth_frames(w, h, nf, seed=100 + s) for s in range(n)]
    batched, solo, errs = [None] * n, [None] * n, []
    os.environ["X264GPU_BATCH"] = str(n)
    ths = [threading.Thread(target=session, args=(w, h, clips[s], opts, batched, s, errs)) for s in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    del os.environ["X264GPU_BATCH"]
    assert not errs, errs
    for s in range(n):
        session(w, h, clips[s], opts, solo, s, errs)
    assert not errs, errs
    TYPE = {1: "I", 2: "i", 3: "P", 4: "R", 5: "B"}
    print(json.dumps({"equal": [batched[s][0] == solo[s][0] for s in range(n)], "sizes": [len(b[0]) for b in batched],
                      "distinct": len({b[0] for b in batched}), "types": "".join(TYPE[t] for t in batched[0][1])}))


main()

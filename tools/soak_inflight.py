#!/usr/bin/env python3
"""Soak of the pictures-in-flight path on the device: random sessions (size, picture count, B structure, references, weightp, direct mode, b-adapt, rate control,
lookahead depth, slices, AQ, keyint / scene cuts) through x264_encoder_encode, once one picture a call (X264GPU_INFLIGHT=0) and once with 2 - 4 pictures in flight;
bytes, picture types, pts / dts and nal_ref_idc of every output must be equal.   Usage: soak_inflight.py SEED CASES [big]   -> one line per mismatch, a summary line"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import host_lib as HL  # noqa: E402
from synth import synth_frames  # noqa: E402

H = HL.H


def session(w, h, frames, opts, inflight):
    os.environ["X264GPU_INFLIGHT"] = str(inflight)
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
    for k, v in opts.items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, (k, v)
    p.b_annexb, p.b_repeat_headers = 1, 1
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_, opts
    depth = H.x264host_pictures_in_flight(h_)
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
        pic.i_pts = i
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), C.byref(pic), C.byref(out))
        assert size >= 0
        take(size)
    while H.x264_encoder_delayed_frames(h_):
        size = H.x264_encoder_encode(h_, C.byref(nal), C.byref(n), None, C.byref(out))
        assert size > 0
        take(size)
    H.x264_encoder_close(h_)
    H.x264_picture_clean(C.byref(pic))
    return stream, meta, depth


def main():
    seed, cases = int(sys.argv[1]), int(sys.argv[2])
    big = len(sys.argv) > 3
    rng = random.Random(seed)
    bad = ran = 0
    for c in range(cases):
        w = 16 * rng.randint(4, 40 if big else 14)
        h = 16 * rng.randint(3, 24 if big else 10)
        nfr = rng.randint(3, 28)
        opts = {"threads": 1, "bframes": rng.randint(1, 3), "ref": rng.randint(1, 4), "weightp": rng.choice([0, 1, 2]), "b-adapt": rng.choice([0, 0, 1, 2]),
                "b-pyramid": rng.choice(["none", "normal", "normal", "strict"]), "direct": rng.choice(["spatial", "spatial", "temporal"]),
                "keyint": rng.choice([250, 250, 12, 7]), "scenecut": rng.choice([40, 40, 0])}
        if rng.random() < 0.5:
            opts["qp"] = rng.randint(18, 36)
        else:
            opts["crf"] = rng.randint(18, 32)
            opts["rc-lookahead"] = rng.choice([0, 4, 10, 20])
            if rng.random() < 0.3: opts["no-mbtree"] = None
            if rng.random() < 0.3: opts["aq-mode"] = rng.choice([0, 2, 3])
        if rng.random() < 0.25 and h >= 128: opts["slices"] = rng.randint(2, 3)
        if rng.random() < 0.2: opts["trellis"] = rng.choice([0, 2])
        if rng.random() < 0.2: opts["subme"] = rng.choice([5, 6, 8, 9])
        if rng.random() < 0.2: opts["me"] = rng.choice(["dia", "umh"])
        if opts["b-pyramid"] == "strict": opts["b-pyramid"] = "normal"          # (strict: blu-ray's; not offered)
        frames = synth_frames(w, h, nfr, seed=seed * 1000 + c, scene_len=rng.choice([97, 9, 5]))
        try:
            s0, m0, d0 = session(w, h, frames, opts, 0)
            k = rng.choice([2, 3, 4, 4])
            s1, m1, d1 = session(w, h, frames, opts, k)
        except AssertionError as e:
            print("case %d: session failed: %r %r" % (c, opts, e))
            bad += 1
            continue
        ran += d1 > 1
        if s0 != s1 or m0 != m1 or d0 != 1:
            bad += 1
            print("MISMATCH case %d seed %d: %dx%d %d pictures, depth %d, %r" % (c, seed, w, h, nfr, d1, opts))
    print("soak_inflight: %d cases, %d with pictures in flight, %d mismatches" % (cases, ran, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

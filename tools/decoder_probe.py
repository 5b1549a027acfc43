#!/usr/bin/env python3
"""A third-party H.264 decoder as an independent check of this encoder's streams.

The oracle restates libx264 from memory and the only decoder the parity suite has is the builder's own (oracle/h264dec.cpp), so the CABAC / CAVLC
tables and the normative reconstruction are checked against the builder's reading of the standard alone.  This module looks, at RUN TIME, for a
decoder nobody here wrote — the ffmpeg CLI, gst-launch-1.0 with an H.264 decoder element, the rocDecode sample decoder — and, when one is there,
decodes an Annex-B stream coded through x264_encoder_encode() and compares every picture with the encoder's own reconstruction
(x264host_get_recon) byte for byte.  The build image and the GPU pool ship none of them (`found` is then false and nothing else is claimed);
libraries that are present without a CLI to drive them (librocdecode.so, libva.so, libavcodec.so) are listed as seen-but-not-driven.

    python tools/decoder_probe.py            # prints the probe's JSON (needs a GPU for the encode)

Used by bench.py (`decoder_probe` in the bench line) and tests/test_gpu_host.py::test_third_party_decoder_agrees (skips when nothing is found)."""
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def find_decoders():
    """-> {"cli": {name: path}, "libraries": {name: path}}: what the box offers"""
    cli = {}
    for name in ("ffmpeg", "gst-launch-1.0", "videodecode", "rocdecdecode"):
        p = shutil.which(name)
        if p:
            cli[name] = p
    for p in glob.glob("/opt/rocm*/share/rocdecode/samples/**/videodecode*", recursive=True) + glob.glob("/opt/rocm*/bin/videodecode*"):
        if os.access(p, os.X_OK) and os.path.isfile(p):
            cli.setdefault("videodecode", p)
    libs = {}
    pats = {"librocdecode": ["/opt/rocm*/lib*/librocdecode.so*"], "libva": ["/usr/lib*/**/libva.so*", "/usr/local/lib*/libva.so*"],
            "libavcodec": ["/usr/lib*/**/libavcodec.so*", "/usr/local/lib*/libavcodec.so*"], "libopenh264": ["/usr/lib*/**/libopenh264.so*"]}
    for name, pp in pats.items():
        hits = sorted(h for p in pp for h in glob.glob(p, recursive=True))
        if hits:
            libs[name] = hits[0]
    return {"cli": cli, "libraries": libs}


def _decode_with(cli, stream_path, w, h, out_path):
    """run one CLI decoder: Annex-B file -> raw I420 frames in display order; -> (ok, what-ran)"""
    if "ffmpeg" in cli:
        cmd = [cli["ffmpeg"], "-hide_banner", "-loglevel", "error", "-y", "-f", "h264", "-i", stream_path, "-f", "rawvideo", "-pix_fmt", "yuv420p", "-vsync", "0", out_path]
        r = subprocess.run(cmd, capture_output=True, timeout=120)
        if r.returncode == 0 and os.path.exists(out_path):
            return True, "ffmpeg"
    if "gst-launch-1.0" in cli:
        for dec in ("avdec_h264", "openh264dec", "vah264dec ! vapostproc", "vaapih264dec ! vaapipostproc"):
            cmd = f"{cli['gst-launch-1.0']} -q filesrc location={stream_path} ! h264parse ! {dec} ! videoconvert ! video/x-raw,format=I420 ! filesink location={out_path}"
            r = subprocess.run(cmd.split(), capture_output=True, timeout=120)
            if r.returncode == 0 and os.path.exists(out_path) and os.path.getsize(out_path) > 0:
                return True, "gst-launch-1.0 " + dec.split()[0]
    if "videodecode" in cli:
        # rocDecode's sample: -i input -o output (NV12 / I420 of the decoded surfaces, display order)
        r = subprocess.run([cli["videodecode"], "-i", stream_path, "-o", out_path], capture_output=True, timeout=120)
        if r.returncode == 0 and os.path.exists(out_path) and os.path.getsize(out_path) > 0:
            return True, "rocDecode videodecode"
    return False, None


def encode_session(w=352, h=288, nfr=12, opts=None):
    """preset medium through x264_encoder_encode(): -> (Annex-B bytes, {pts: reconstruction I420})"""
    import numpy as np
    from x264vfw_amd import host_api as HL
    from x264vfw_amd.synth import synth_frames
    H = HL.H
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
    p.i_width, p.i_height, p.i_csp = w, h, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
    for k, v in dict({"qp": "24", "keyint": "250", "threads": "1"}, **(opts or {})).items():
        assert H.x264_param_parse(C.byref(p), k.encode(), None if v is None else str(v).encode()) == 0, k
    p.b_annexb, p.b_repeat_headers = 1, 1
    h_ = H.x264_encoder_open_157(C.byref(p))
    assert h_
    pic, out = HL.Picture(), HL.Picture()
    assert H.x264_picture_alloc(C.byref(pic), HL.X264_CSP_I420, w, h) == 0
    nal, nn = C.POINTER(HL.Nal)(), C.c_int()
    planes = [(w * h, 0), (w * h // 4, w * h), (w * h // 4, w * h * 5 // 4)]
    frames = synth_frames(w, h, nfr, seed=0x1264)
    stream, recons = b"", {}

    def take(size):
        nonlocal stream
        if size <= 0:
            return
        stream += C.string_at(nal[0].p_payload, size)
        rec = np.empty(w * h * 3 // 2, dtype=np.uint8)
        assert H.x264host_get_recon(h_, rec.ctypes.data) == 0
        recons[int(out.i_pts)] = rec

    for i in range(nfr):
        f = frames[i]
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
    assert len(recons) == nfr
    return stream, recons


def probe(w=352, h=288, nfr=12, opts=None):
    """-> {"found": bool, "decoder": str | None, "equal": bool | None, "pictures": n, "seen_not_driven": {...}}; never raises"""
    import numpy as np
    found = find_decoders()
    res = {"found": bool(found["cli"]), "decoder": None, "equal": None, "pictures": 0, "seen_not_driven": found["libraries"],
           "what": "an Annex-B stream of preset medium through x264_encoder_encode(), decoded by a third-party decoder found on this box at run time and compared "
                   "byte for byte with x264host_get_recon of every picture (display order); found = false: no decoder CLI on the box, nothing checked"}
    if not found["cli"]:
        return res
    try:
        stream, recons = encode_session(w, h, nfr, opts)
        with tempfile.TemporaryDirectory() as td:
            sp, op = os.path.join(td, "probe.264"), os.path.join(td, "probe.yuv")
            open(sp, "wb").write(stream)
            ok, which = _decode_with(found["cli"], sp, w, h, op)
            res["decoder"] = which
            if not ok:
                res["error"] = "a decoder CLI is present but did not decode the stream"
                res["equal"] = False
                return res
            raw = np.fromfile(op, dtype=np.uint8)
        n = w * h * 3 // 2
        res["pictures"] = int(raw.size // n)
        eq = raw.size == n * nfr
        for i in range(min(nfr, raw.size // n)):
            eq = eq and bool(np.array_equal(raw[i * n:(i + 1) * n], recons[i]))
        res["equal"] = bool(eq)
    except Exception as e:          # the probe never fails its caller
        res["error"] = repr(e)[:300]
    return res


if __name__ == "__main__":
    print(json.dumps(probe()))

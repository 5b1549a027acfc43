"""-m gpu: the reference's ICM message sequence replayed end to end against the MI355X
path: FRAMES_INFO -> GET_FORMAT -> QUERY -> BEGIN -> COMPRESS x N -> END -> CLOSE."""
import ctypes as C

import numpy as np
import pytest

import host_lib as V
import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.gpu
D = V.H.DriverProc


@pytest.mark.parametrize("four,cmdline", [(b"I420", b"--keyint 5 --no-scenecut --profile baseline"), (b"YV12", b"--keyint 250 --no-deblock --subme 4")])
def test_icm_compress_sequence(gpu, four, cmdline):
    w, h, nfr = 176, 144, 7
    frames = synth_frames(w, h, nfr, seed=1234)
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.i_encoding_type, cfg.i_qp, cfg.extra_cmdline = 1, 27, cmdline + b" --bframes 0 --weightp 0"        # single pass CQP (codec.c:1498-1502)
    assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == n
    inb, outb = V.bmi(w, h, four), V.BITMAPINFO()
    icf = V.ICCOMPRESSFRAMES(lFrameCount=nfr, dwRate=25, dwScale=1)
    assert D(cid, None, V.ICM_COMPRESS_FRAMES_INFO, V.addr(icf), C.sizeof(icf)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    cap = outb.bmiHeader.biSizeImage
    buf = C.create_string_buffer(cap)
    stream, keys = b"", []
    for f in frames:
        y, u, v = f[:w * h], f[w * h:w * h * 5 // 4], f[w * h * 5 // 4:]
        src = np.concatenate([y, v, u]) if four == b"YV12" else f           # YV12 = V plane first (csp.c:412)
        src = np.ascontiguousarray(src)
        flags = V.DWORD(0xdead)
        outb.bmiHeader.biSizeImage = cap
        icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                           lpInput=src.ctypes.data, lpdwFlags=C.pointer(flags))
        assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK
        size = outb.bmiHeader.biSizeImage
        assert 0 < size <= cap
        stream += buf.raw[:size]
        keys.append(flags.value)
    assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
    assert D(cid, None, V.DRV_CLOSE, 0, 0) == 1
    keyint = 5 if b"keyint 5" in cmdline else 250
    assert keys == [V.AVIIF_KEYFRAME if i % keyint == 0 else 0 for i in range(nfr)]
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    for i in range(nfr):
        assert O.L is not None
        from synth import psnr
        assert psnr(dec[i][:w * h], frames[i][:w * h]) > 33.0, f"frame {i} decodes to something unlike the input"


def test_small_output_buffer_is_an_error(gpu):
    w, h = 64, 48
    f = synth_frames(w, h, 1, seed=2)[0]
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.b_zerolatency = 1                                                    # the first call has to produce a frame
    assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == n
    inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
    D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb))
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    outb.bmiHeader.biSizeImage = 16                                          # codec.c:1710-1718 guard
    buf = C.create_string_buffer(16)
    flags = V.DWORD()
    icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                       lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
    assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), 0) == V.ICERR_ERROR
    assert b"output frame buffer too small" in V.H.x264vfw_shim_log(cid)
    assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), 0) == V.ICERR_ERROR     # sticky b_encoder_error
    D(cid, None, V.DRV_CLOSE, 0, 0)


def test_default_session_delays_output_by_the_lookahead(gpu, tmp_path):
    """the driver's defaults (single pass CRF, preset medium: macroblock-tree over rc-lookahead 40): ICM_COMPRESS hands back empty
    frames while the lookahead fills, warns once about the frames a VfW host will lose (codec.c:1798-1807), and with file output
    compress_end flushes every picture (codec.c:1842-1856)."""
    w, h, nfr, look = 96, 80, 9, 5
    frames = synth_frames(w, h, nfr, seed=8)

    def run(cmdline):
        ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
        cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
        n = D(cid, None, V.ICM_GETSTATE, 0, 0)
        cfg = V.VfwConfig()
        D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
        assert cfg.i_encoding_type == 2 and cfg.b_zerolatency == 0          # config.c defaults: single pass ratefactor-based (CRF)
        cfg.extra_cmdline = cmdline + b" --bframes 0 --weightp 0"          # (the macroblock-tree session: B sessions run without it so far)
        D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
        inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
        assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
        assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
        cap = outb.bmiHeader.biSizeImage
        buf = C.create_string_buffer(cap)
        stream, sizes, keys = b"", [], []
        for f in frames:
            flags = V.DWORD(0xdead)
            outb.bmiHeader.biSizeImage = cap
            icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                               lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
            assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK
            sizes.append(outb.bmiHeader.biSizeImage)
            keys.append(flags.value)
            stream += buf.raw[:outb.bmiHeader.biSizeImage]
        log = V.H.x264vfw_shim_log(cid)
        assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
        D(cid, None, V.DRV_CLOSE, 0, 0)
        return stream, sizes, keys, log

    opts = b"--rc-lookahead %d --keyint 250 " % look
    stream, sizes, keys, log = run(opts)
    delay = look + 1                                                         # + 1: the GPU stage of a picture overlaps the entropy coding of the one before
    assert sizes[:delay] == [0] * delay and all(s > 0 for s in sizes[delay:])
    assert keys[:delay + 1] == [0] * delay + [V.AVIIF_KEYFRAME] and log.count(b"Few frames probably would be lost") == 1
    dec = O.h264_decode(stream, nfr - delay, w, h)                           # the tail stays in the lookahead: the loss the log warns of
    assert len(dec) == nfr - delay
    path = tmp_path / "d.h264"
    to_file, sizes, keys, log = run(opts + b"--output " + str(path).encode())
    assert to_file == b"" and b"Few frames" not in log
    data = path.read_bytes()
    assert data.startswith(stream)                                           # same pictures, then the flushed tail
    from synth import psnr
    full = O.h264_decode(data, nfr, w, h)
    ps = [psnr(full[i][:w * h], frames[i][:w * h]) for i in range(len(full))]
    assert len(full) == nfr and min(ps) > 30.0, ps
    # zero latency (config.c / codec.c:1439 tune zerolatency): rc-lookahead 0, a frame per call
    stream0, sizes0, _, _ = run(b"--tune zerolatency --keyint 250")
    assert all(s > 0 for s in sizes0)


@pytest.mark.parametrize("fmt", ["YUY2", "UYVY", "YV16", "YV24", "RGB24", "RGB32", "RGB32_topdown"])
def test_icm_compress_native_colourspaces(gpu, fmt):
    """Packed / RGB / 4:2:2 / 4:4:4 inputs (codec.c:187-231): the shell uploads the native frame and converts on the device;
    the stream must equal the one produced from the oracle-converted I420 picture (csp.c restated in oracle/csp.c)."""
    w, h, nfr = 96, 80, 3
    rng = np.random.default_rng(7)
    CSP = O.CSP
    four, bits, height, csp = {"YUY2": (b"YUY2", 16, h, CSP["YUYV"]), "UYVY": (b"UYVY", 16, h, CSP["UYVY"]), "YV16": (b"YV16", 16, h, CSP["YV16"]),
                               "YV24": (b"YV24", 24, h, CSP["YV24"]), "RGB24": (None, 24, h, CSP["BGR"] | CSP["VFLIP"]),
                               "RGB32": (None, 32, h, CSP["BGRA"] | CSP["VFLIP"]), "RGB32_topdown": (None, 32, -h, CSP["BGRA"])}[fmt]
    # smooth-ish content so the encode is meaningful: low-pass noise
    base = rng.integers(0, 256, (nfr, O.csp_img_fill(csp, w, h)[0]), dtype=np.uint8)
    raws = [np.ascontiguousarray(b) for b in base]

    def run(frames, fourcc, bitcount, hh):
        ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
        cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
        n = D(cid, None, V.ICM_GETSTATE, 0, 0)
        cfg = V.VfwConfig()
        D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
        cfg.i_encoding_type, cfg.i_qp, cfg.extra_cmdline = 1, 30, b"--keyint 250 --colormatrix bt709 --range pc --bframes 0 --weightp 0"
        assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == n
        inb, outb = V.bmi(w, hh, fourcc if fourcc else b"\x00\x00\x00\x00"), V.BITMAPINFO()
        inb.bmiHeader.biBitCount = bitcount
        assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
        assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), V.addr(outb)) == V.ICERR_OK
        assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
        cap = outb.bmiHeader.biSizeImage
        buf = C.create_string_buffer(cap)
        stream = b""
        for f in frames:
            flags = V.DWORD(0)
            outb.bmiHeader.biSizeImage = cap
            icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                               lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
            assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
            stream += buf.raw[:outb.bmiHeader.biSizeImage]
        assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
        assert D(cid, None, V.DRV_CLOSE, 0, 0) == 1
        return stream

    native = run(raws, four, bits, height)
    converted = [np.ascontiguousarray(O.csp_to_i420(r, csp, w, h, 1, 1)) for r in raws]       # bt709, pc range as configured
    via_i420 = run(converted, b"I420", 12, h)
    assert native == via_i420
    assert len(O.h264_decode(native, nfr, w, h)) == nfr


def test_unsupported_input_is_badformat(gpu):
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    inb, outb = V.bmi(64, 48, b"NV12"), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), 0) == V.ICERR_BADFORMAT
    inb = V.bmi(64, 48, b"\x00\x00\x00\x00")
    inb.bmiHeader.biBitCount = 16                                            # RGB555/565 are not driver inputs (codec.c:218-226)
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_BADFORMAT
    D(cid, None, V.DRV_CLOSE, 0, 0)


def test_raw_file_output(gpu, tmp_path):
    """--output file.h264 (select_output -> raw_output, codec.c:1111-1164 / output/raw.c): frames go to the file as Annex-B,
    ICM_COMPRESS hands the host application empty frames, and the file decodes to the same pictures as the buffer path."""
    w, h, nfr = 96, 80, 4
    frames = synth_frames(w, h, nfr, seed=5)
    path = tmp_path / "out.h264"

    def run(cmdline):
        ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
        cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
        n = D(cid, None, V.ICM_GETSTATE, 0, 0)
        cfg = V.VfwConfig()
        D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
        cfg.i_encoding_type, cfg.i_qp, cfg.extra_cmdline = 1, 28, cmdline + b" --bframes 0 --weightp 0"
        D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
        inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
        assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
        rc = D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb))
        if rc != V.ICERR_OK:
            log = V.H.x264vfw_shim_log(cid)
            D(cid, None, V.DRV_CLOSE, 0, 0)
            return None, log
        cap = outb.bmiHeader.biSizeImage
        buf = C.create_string_buffer(cap)
        stream, sizes = b"", []
        for f in frames:
            flags = V.DWORD(0)
            outb.bmiHeader.biSizeImage = cap
            icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                               lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
            assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK
            sizes.append(outb.bmiHeader.biSizeImage)
            stream += buf.raw[:outb.bmiHeader.biSizeImage]
        assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
        D(cid, None, V.DRV_CLOSE, 0, 0)
        return stream, sizes

    direct, sizes = run(b"--keyint 250")
    assert all(s > 0 for s in sizes)
    to_file, sizes = run(b"--keyint 250 --output " + str(path).encode())
    assert to_file == b"" and sizes == [0] * nfr                             # codec.c:1708-1721: nothing reaches the VfW buffer
    assert path.read_bytes() == direct
    # GOP-parallel coding (--threads 3): frames reach the file late and compress_end flushes the rest (codec.c:1842-1856);
    # the file is byte-identical to the serial one
    par_path = tmp_path / "par.h264"
    run(b"--keyint 2 --min-keyint 2 --no-scenecut --output " + str(path).encode())
    serial_k2 = path.read_bytes()
    to_file, sizes = run(b"--keyint 2 --min-keyint 2 --no-scenecut --threads 3 --output " + str(par_path).encode())
    assert to_file == b"" and par_path.read_bytes() == serial_k2
    # without a file the VfW buffer cannot take late frames: threads falls back to 1 (same stream, no delay)
    direct_t, sizes = run(b"--keyint 250 --threads 4")
    assert direct_t == direct and all(s > 0 for s in sizes)
    none, log = run(b"--output " + str(tmp_path / "x.avi").encode())
    assert none is None and b"output support" in log                         # avi needs libavformat: not built in
    # containers (next-row f3): length-prefixed NALs, parameter sets once in the avcC record; same slices as the raw stream
    import container_parse as CP
    import oracle_lib as O
    raw_nals = [n for n in direct.split(b"\0\0\1")[1:]]
    raw_slices = [n.rstrip(b"\0") if i + 1 < len(raw_nals) else n for i, n in enumerate(raw_nals)]
    raw_slices = [n for n in raw_slices if (n[0] & 31) in (1, 5)]
    for ext in ("mkv", "flv", "mp4"):
        cpath = tmp_path / ("c." + ext)
        to_file, sizes = run(b"--keyint 250 --output " + str(cpath).encode())
        assert to_file == b"" and sizes == [0] * nfr
        c = CP.mkv_read(cpath.read_bytes()) if ext == "mkv" else CP.flv_read(cpath.read_bytes()) if ext == "flv" else CP.mp4_read(cpath.read_bytes())
        if ext == "mp4":
            c["frames"] = c["samples"]
            assert (c["media_timescale"], c["deltas"], [f["key"] for f in c["frames"]]) == (25, [1] * nfr, [True] + [False] * (nfr - 1))
            if __import__("os").path.exists(O.LSMASH_REF):                   # and through the reference tree's L-SMASH
                info, smp, data = O.lsmash_read_mp4(cpath)
                assert info.n_samples == nfr and data == [f["data"] for f in c["frames"]] and [x.sync for x in smp] == [1] + [0] * (nfr - 1)
        a = CP.avcc_read(c["avcc"])
        slices = [CP.length_prefixed_nals(f["data"])[-1] for f in c["frames"]]
        assert slices == raw_slices, ext
        es = b"\0\0\0\1" + a["sps"] + b"\0\0\0\1" + a["pps"] + b"".join(b"\0\0\0\1" + n for n in slices)
        assert len(O.h264_decode(es, nfr, w, h)) == nfr
    # GOP-parallel coding into a container: frames reach the muxer late but in order, with their own timestamps
    run(b"--keyint 2 --min-keyint 2 --no-scenecut --output " + str(tmp_path / "s.mkv").encode())
    run(b"--keyint 2 --min-keyint 2 --no-scenecut --threads 3 --output " + str(tmp_path / "p.mkv").encode())
    ms, mp = CP.mkv_read((tmp_path / "s.mkv").read_bytes()), CP.mkv_read((tmp_path / "p.mkv").read_bytes())
    assert [(f["timecode"], f["key"], f["data"]) for f in mp["frames"]] == [(f["timecode"], f["key"], f["data"]) for f in ms["frames"]]
    assert [f["timecode"] for f in ms["frames"]] == sorted(f["timecode"] for f in ms["frames"]) and len(ms["frames"]) == nfr


def _session(cfg_edit, w=64, h=48, nfr=3, fmt=b"I420"):
    """DRV_OPEN -> SETSTATE(cfg) -> BEGIN -> COMPRESS x nfr -> END -> CLOSE; -> (begin rc, stream, sizes, log)"""
    frames = synth_frames(w, h, nfr, seed=3)
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg_edit(cfg)
    assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == n
    inb, outb = V.bmi(w, h, fmt), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    rc = D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb))
    stream, sizes = b"", []
    if rc == V.ICERR_OK:
        cap = outb.bmiHeader.biSizeImage
        buf = C.create_string_buffer(cap)
        for f in frames:
            flags = V.DWORD(0)
            outb.bmiHeader.biSizeImage = cap
            icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                               lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
            assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
            sizes.append(outb.bmiHeader.biSizeImage)
            stream += buf.raw[:outb.bmiHeader.biSizeImage]
    log = V.H.x264vfw_shim_log(cid)
    D(cid, None, V.ICM_COMPRESS_END, 0, 0)
    D(cid, None, V.DRV_CLOSE, 0, 0)
    return rc, stream, sizes, log, frames


def test_every_config_dialog_choice_opens_a_session(gpu):
    """the configuration dialog's whole range (config.c: 10 presets, 7 tunings, 4 profiles, the level list, fast decode / zero latency,
    the five rate-control modes): every choice either runs — tools this round does not have are stepped down to the nearest built one
    with a log line, never silently — or is refused with a message; whatever comes out decodes"""
    def one(edit, what):
        def full(cfg):
            cfg.b_zerolatency = 1                                            # a frame per call, so the output can be checked
            edit(cfg)
        rc, stream, sizes, log, frames = _session(full)
        assert rc == V.ICERR_OK, (what, log)
        assert all(s > 0 for s in sizes), (what, sizes, log)
        dec = O.h264_decode(stream, len(frames), 64, 48)
        assert len(dec) == len(frames), what
    for preset in range(10):
        one(lambda c: setattr(c, "i_preset", preset), f"preset {preset}")
    for tuning in range(7):
        one(lambda c: setattr(c, "i_tuning", tuning), f"tuning {tuning}")
    for profile in range(4):
        one(lambda c: setattr(c, "i_profile", profile), f"profile {profile}")
    for level in range(0, 18):
        one(lambda c: setattr(c, "i_level", level), f"level {level}")
    one(lambda c: setattr(c, "b_fastdecode", 1), "fastdecode")
    for enc_type in (1, 2, 3):
        def e(c):
            c.i_encoding_type, c.i_qp, c.i_rf_constant, c.i_passbitrate = enc_type, 26, 230, 400
        one(e, f"encoding type {enc_type}")
    # lossless (qp 0 + transform bypass) needs tools outside this subset: refused loudly or stepped down with a log line
    rc, stream, sizes, log, frames = _session(lambda c: (setattr(c, "i_encoding_type", 0), setattr(c, "b_zerolatency", 1)))
    assert rc != V.ICERR_OK or (all(s > 0 for s in sizes) and len(O.h264_decode(stream, len(frames), 64, 48)) == len(frames)), log
    assert rc == V.ICERR_OK or len(log) > 0
    # multipass, first pass (the dialog's default i_pass 1): the statistics file and no stream (codec.c:1519-1524 b_no_output); test_two_pass_through_the_driver
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        st = os.path.join(td, "a.stats").encode()
        rc, stream, sizes, log, frames = _session(lambda c: (setattr(c, "i_encoding_type", 4), setattr(c, "extra_cmdline", b"--stats " + st)))
        assert rc == V.ICERR_OK and not any(sizes), (sizes, log)


def test_driver_defaults_with_b_pictures_to_a_file(gpu, tmp_path):
    """the driver's own defaults untouched — single pass CRF, preset medium: bframes 3, b-adapt 1, b-pyramid, weightb, weightp 2, mbtree — through
    DriverProc with file output: ICM_COMPRESS returns empty frames (the pictures leave in coding order into the file), ICM_COMPRESS_END flushes the
    lookahead and the mini-GOP in flight (codec.c:1842-1856), the file holds every picture"""
    w, h, nfr = 96, 80, 17
    frames = synth_frames(w, h, nfr, seed=5)
    path = tmp_path / "b.h264"
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.extra_cmdline = b"--rc-lookahead 6 --output " + str(path).encode()
    D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
    inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    cap = outb.bmiHeader.biSizeImage
    buf = C.create_string_buffer(cap)
    for f in frames:
        flags = V.DWORD(0)
        outb.bmiHeader.biSizeImage = cap
        icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                           lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
        assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    log = V.H.x264vfw_shim_log(cid)
    assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
    D(cid, None, V.DRV_CLOSE, 0, 0)
    assert b"bframes 0" not in log and b"mbtree 0" not in log, log
    data = path.read_bytes()
    dec = O.h264_decode(data, nfr, w, h)
    pocs = O.h264_last_pocs()
    assert len(dec) == nfr
    from synth import psnr
    # closed GOPs: POC restarts at IDR pictures; display index = pictures before the IDR + poc / 2
    base, disp, last = 0, [], -1
    for p in pocs:
        if p == 0 and disp:
            base = max(disp) + 1
        disp.append(base + p // 2)
    assert sorted(disp) == list(range(nfr))
    for d, i in zip(dec, disp):
        assert psnr(d[:w * h], frames[i][:w * h]) > 27.0


def test_virtualdub_hack_carries_b_pictures_through_the_vfw_buffer(gpu):
    """--vd-hack (X264VFW_USE_VIRTUALDUB_HACK, codec.c:1809-1820): a call that hands in a frame while the lookahead / the mini-GOP hold the
    pictures back answers with a one-byte 0x7f drop frame under the XVID fourcc; VirtualDub then keeps calling after the last frame
    (ICM_COMPRESS_FRAMES_INFO told the driver the count: i_frame_remain, codec.c:1756-1790) until every picture has come back.  The real frames,
    in the order they left, are the stream: B pictures survive VfW's one-in-one-out protocol"""
    w, h, nfr = 96, 80, 15
    frames = synth_frames(w, h, nfr, seed=6)
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.b_vd_hack = 1                                                          # the dialog's box (codec.c:1410); --vd-hack on the command line does the same
    cfg.extra_cmdline = b"--rc-lookahead 5"
    D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
    inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    fi = V.ICCOMPRESSFRAMES(lFrameCount=nfr, dwRate=25, dwScale=1)
    assert D(cid, None, V.ICM_COMPRESS_FRAMES_INFO, V.addr(fi), C.sizeof(fi)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    fourcc = outb.bmiHeader.biCompression
    cap = outb.bmiHeader.biSizeImage
    buf = C.create_string_buffer(cap)
    stream, drops, keyflags = b"", 0, []
    calls = 0
    while len(keyflags) < nfr and calls < 4 * nfr:
        f = frames[min(calls, nfr - 1)]                         # after the last frame VirtualDub repeats its calls: the driver ignores the input then
        flags = V.DWORD(0)
        outb.bmiHeader.biSizeImage = cap
        icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                           lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
        assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
        calls += 1
        size = outb.bmiHeader.biSizeImage
        if size == 1 and buf.raw[0] == 0x7f:
            assert outb.bmiHeader.biCompression == V.fourcc(b"XVID") and flags.value == 0
            drops += 1
        else:
            assert size > 1 and outb.bmiHeader.biCompression == fourcc
            stream += buf.raw[:size]
            keyflags.append(flags.value)
    log = V.H.x264vfw_shim_log(cid)
    assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
    D(cid, None, V.DRV_CLOSE, 0, 0)
    assert len(keyflags) == nfr and drops == calls - nfr and drops >= 5, (calls, drops)
    assert b"Few frames probably would be lost" not in log and b"bframes 0" not in log, log
    assert keyflags[0] == V.AVIIF_KEYFRAME
    dec = O.h264_decode(stream, nfr, w, h)
    assert len(dec) == nfr
    pocs = O.h264_last_pocs()
    assert sorted(p // 2 for p in pocs) == list(range(nfr)) and [p // 2 for p in pocs] != list(range(nfr))        # B pictures: coding order != display order
    from synth import psnr
    for d, p in zip(dec, pocs):
        assert psnr(d[:w * h], frames[p // 2][:w * h]) > 27.0


@pytest.mark.parametrize("ext", ["mp4", "mkv", "flv"])
def test_b_pictures_into_the_containers(gpu, tmp_path, ext):
    """the driver's default session (B pictures, b-pyramid) through DriverProc into the reference's container outputs (output/mp4_lsmash.c,
    matroska.c, flv.c as restated in host/muxers.cpp): the samples lie in coding order with decode times that never run backwards, the presentation
    times are a permutation of the frame times (mp4: composition offsets, read back by the reference tree's L-SMASH too; mkv: block timecodes,
    B pictures that nothing references marked discardable, output/matroska.c:199-202; flv: CompositionTime), and the elementary stream decodes to
    the source"""
    import container_parse as CP
    w, h, nfr = 96, 80, 13
    frames = synth_frames(w, h, nfr, seed=8)
    path = tmp_path / ("b." + ext)
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.extra_cmdline = b"--rc-lookahead 4 --b-adapt 0 --scenecut 0 --output " + str(path).encode()
    D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
    inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    fi = V.ICCOMPRESSFRAMES(lFrameCount=nfr, dwRate=25, dwScale=1)
    assert D(cid, None, V.ICM_COMPRESS_FRAMES_INFO, V.addr(fi), C.sizeof(fi)) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    cap = outb.bmiHeader.biSizeImage
    buf = C.create_string_buffer(cap)
    for f in frames:
        flags = V.DWORD(0)
        outb.bmiHeader.biSizeImage = cap
        icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                           lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
        assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
    assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
    D(cid, None, V.DRV_CLOSE, 0, 0)
    raw = path.read_bytes()
    c = CP.mkv_read(raw) if ext == "mkv" else CP.flv_read(raw) if ext == "flv" else CP.mp4_read(raw)
    fr = c["samples"] if ext == "mp4" else c["frames"]
    assert len(fr) == nfr
    a = CP.avcc_read(c["avcc"])
    es = b"\0\0\0\1" + a["sps"] + b"\0\0\0\1" + a["pps"] + b"".join(b"\0\0\0\1" + nal for f in fr for nal in CP.length_prefixed_nals(f["data"]))
    dec = O.h264_decode(es, nfr, w, h)
    pocs = O.h264_last_pocs()
    assert len(dec) == nfr and sorted(p // 2 for p in pocs) == list(range(nfr)) and [p // 2 for p in pocs] != list(range(nfr))
    disp = [p // 2 for p in pocs]                                              # display index of every sample, in file (= coding) order
    if ext == "mp4":
        dts, cts = [x["dts"] for x in fr], [x["cts"] for x in fr]
        assert dts == sorted(dts) and len(set(dts)) == nfr and all(ct >= dt for ct, dt in zip(cts, dts))
        assert [ct - min(cts) for ct in cts] == [d * (cts[1] - cts[0]) // (disp[1] - disp[0]) for d in disp]        # presentation order = display order
        if __import__("os").path.exists(O.LSMASH_REF):
            info, smp, data = O.lsmash_read_mp4(path)
            assert info.n_samples == nfr and [x.cts - x.dts for x in smp] == [ct - dt for ct, dt in zip(cts, dts)]
    elif ext == "mkv":
        tcs = [x["timecode"] for x in fr]
        assert sorted(tcs) == [tcs[disp.index(i)] for i in range(nfr)] and len(set(tcs)) == nfr                 # block timecodes are presentation times
        assert any(x["discardable"] for x in fr) and fr[0]["key"] and not fr[0]["discardable"]
    else:
        dts = [x["dts"] for x in fr]
        assert dts == sorted(dts)
        pts = [x["dts"] + x["cts_offset"] for x in fr]
        assert sorted(pts) == [pts[disp.index(i)] for i in range(nfr)] and all(x["cts_offset"] >= 0 for x in fr)
    from synth import psnr
    for d, i in zip(dec, disp):
        assert psnr(d[:w * h], frames[i][:w * h]) > 27.0


def test_two_pass_through_the_driver(gpu, tmp_path):
    """encoding type 4 of the config dialog (codec.c:1516-1533): pass 1 returns no stream and writes the statistics file, pass 2 reads it and hits the
    requested bitrate within 5 % (x264 ratecontrol.c init_pass2 restated in host/encoder.cpp)"""
    w, h, nfr, kbps = 176, 144, 240, 300
    frames = synth_frames(w, h, nfr, seed=11, scene_len=53)
    stats = tmp_path / "vfw.stats"

    def run(i_pass, out, enc_type=4, **fields):
        ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
        cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
        n = D(cid, None, V.ICM_GETSTATE, 0, 0)
        cfg = V.VfwConfig()
        D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
        assert (cfg.b_fast1pass, cfg.b_createstats, cfg.b_updatestats) == (0, 0, 1)          # config.c:114-116
        cfg.i_encoding_type, cfg.i_passbitrate, cfg.i_pass, cfg.i_log_level = enc_type, kbps, i_pass, 3
        for k, v in fields.items():
            setattr(cfg, k, v)
        # (the statistics file and the output file as the dialog stores them: CONFIG.stats, CONFIG.i_output_mode / output_file — codec.c:1447,1537-1545)
        cfg.stats, cfg.i_output_mode, cfg.output_file = str(stats).encode(), 1, str(out).encode()
        cfg.extra_cmdline = b"--keyint 40 --rc-lookahead 8"
        D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n)
        inb, outb = V.bmi(w, h, b"I420"), V.BITMAPINFO()
        assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
        assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
        cap = outb.bmiHeader.biSizeImage
        buf = C.create_string_buffer(cap)
        for f in frames:
            flags = V.DWORD(0)
            outb.bmiHeader.biSizeImage = cap
            icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                               lpInput=f.ctypes.data, lpdwFlags=C.pointer(flags))
            assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), C.sizeof(icc)) == V.ICERR_OK, V.H.x264vfw_shim_log(cid)
        log = V.H.x264vfw_shim_log(cid)
        assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
        D(cid, None, V.DRV_CLOSE, 0, 0)
        return log

    run(1, tmp_path / "p1.h264")
    assert stats.exists() and not (tmp_path / "p1.h264").exists()                    # pass 1: b_no_output
    lines = [ln for ln in stats.read_text().splitlines() if not ln.startswith("#")]
    assert len(lines) == nfr
    log = run(2, tmp_path / "p2.h264")
    assert b"planned from the first pass" in log, log
    data = (tmp_path / "p2.h264").read_bytes()
    rate = len(data) * 8 / (nfr / 25.0) / 1000.0
    assert abs(rate / kbps - 1.0) < 0.05, rate
    assert len(O.h264_decode(data, nfr, w, h)) == nfr
    # the dialog's "update stats" box (x264vfw.h:140, default on: codec.c:1527): pass N wrote its own lines over the first pass'; unticked, the file stays
    lines2 = [ln for ln in stats.read_text().splitlines() if not ln.startswith("#")]
    assert len(lines2) == nfr and lines2 != lines
    run(2, tmp_path / "p3.h264", b_updatestats=0)
    assert [ln for ln in stats.read_text().splitlines() if not ln.startswith("#")] == lines2
    # "create stats" (x264vfw.h:139, codec.c:1495-1513): a single-pass session writes the file too — and returns its stream
    stats.unlink()
    run(1, tmp_path / "p4.h264", enc_type=2, b_createstats=1)
    assert (tmp_path / "p4.h264").stat().st_size > 0 and len([ln for ln in stats.read_text().splitlines() if not ln.startswith("#")]) == nfr

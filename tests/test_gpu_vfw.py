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


@pytest.mark.parametrize("four,cmdline", [(b"I420", b"--keyint 5 --profile baseline"), (b"YV12", b"--keyint 250 --no-deblock --subme 4")])
def test_icm_compress_sequence(gpu, four, cmdline):
    w, h, nfr = 176, 144, 7
    frames = synth_frames(w, h, nfr, seed=1234)
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    cfg = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n)
    cfg.i_encoding_type, cfg.i_qp, cfg.extra_cmdline = 1, 27, cmdline        # single pass CQP (codec.c:1498-1502)
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

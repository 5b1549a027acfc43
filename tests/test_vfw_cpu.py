"""CPU: boundary B2 — the ICM message protocol of driverproc.c:89-301 replayed against the Linux re-host
(format negotiation, state blob, error conventions).  The compress data path itself needs a GPU
(tests/test_gpu_vfw.py); here BEGIN must fail loudly and stickily instead of falling back."""
import ctypes as C

import pytest

import host_lib as V

D = V.H.DriverProc


def open_codec():
    ico = V.ICOPEN(fccType=V.fourcc(b"vidc"))
    cid = D(0, None, V.DRV_OPEN, 0, V.addr(ico))
    assert cid and ico.dwError == V.ICERR_OK
    return cid


def test_driver_lifecycle_and_info():
    assert D(0, None, V.DRV_LOAD, 0, 0) == 1 and D(0, None, V.DRV_FREE, 0, 0) == 1
    bad = V.ICOPEN(fccType=V.fourcc(b"audc"))
    assert D(0, None, V.DRV_OPEN, 0, V.addr(bad)) == 0                      # not a video codec request (driverproc.c:109)
    cid = open_codec()
    assert D(cid, None, V.DRV_QUERYCONFIGURE, 0, 0) == 0 and D(cid, None, V.DRV_CONFIGURE, 0, 0) == 0
    info = V.ICINFO()
    assert D(cid, None, V.ICM_GETINFO, V.addr(info), C.sizeof(info)) == C.sizeof(info)
    assert info.fccHandler == V.fourcc(b"X264") and info.fccType == V.fourcc(b"vidc") and info.dwFlags == (0x8 | 0x20)
    assert "".join(map(chr, info.szName)).rstrip("\0") == "x264vfw"
    assert D(cid, None, V.ICM_GETINFO, V.addr(info), 8) == 0               # buffer too small
    assert D(cid, None, V.ICM_DECOMPRESS_QUERY, 0, 0) == V.ICERR_UNSUPPORTED  # decoder is out of scope -> default branch
    assert D(cid, None, V.DRV_USER + 0x999, 0, 0) == V.ICERR_UNSUPPORTED
    assert D(cid, None, V.DRV_CLOSE, 0, 0) == 1


def test_state_blob():
    cid = open_codec()
    n = D(cid, None, V.ICM_GETSTATE, 0, 0)
    assert n == C.sizeof(V.VfwConfig)
    cfg = V.VfwConfig()
    assert D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n - 1) == V.ICERR_BADSIZE
    assert D(cid, None, V.ICM_GETSTATE, V.addr(cfg), n) == V.ICERR_OK
    assert (cfg.i_format_version, cfg.i_preset, cfg.i_encoding_type, cfg.i_rf_constant, cfg.i_qp) == (4, 5, 2, 230, 23)   # Appendix A defaults
    # ... and the rest of the reference's CONFIG (x264vfw.h:121-167) with config.c:96-143's defaults: the blob carries every field the dialog edits
    assert (cfg.i_colorspace, cfg.b_fast1pass, cfg.b_createstats, cfg.b_updatestats, cfg.i_output_mode, cfg.b_vd_hack, cfg.b_disable_decoder) == (0, 0, 0, 1, 0, 0, 0)
    assert cfg.stats == b"./x264.stats" and cfg.output_file == b"" and (cfg.i_sar_width, cfg.i_sar_height, cfg.i_log_level) == (1, 1, 2)
    cfg.i_encoding_type, cfg.i_qp = 1, 30
    assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == n
    cfg2 = V.VfwConfig()
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg2), n)
    assert cfg2.i_qp == 30
    cfg.i_format_version = 3
    assert D(cid, None, V.ICM_SETSTATE, V.addr(cfg), n) == 0               # wrong version is refused (driverproc.c:174)
    assert D(cid, None, V.ICM_SETSTATE, 0, 0) == 0                         # NULL resets to defaults
    D(cid, None, V.ICM_GETSTATE, V.addr(cfg2), n)
    assert cfg2.i_qp == 23
    D(cid, None, V.DRV_CLOSE, 0, 0)


def test_format_negotiation():
    cid = open_codec()
    inb, outb = V.bmi(1920, 1080, b"I420"), V.BITMAPINFO()
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), 0) == C.sizeof(V.BITMAPINFOHEADER)
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    oh = outb.bmiHeader
    assert (oh.biWidth, oh.biHeight, oh.biCompression, oh.biBitCount) == (1920, 1080, V.fourcc(b"H264"), 24)
    assert oh.biSizeImage == 6270976 == D(cid, None, V.ICM_COMPRESS_GET_SIZE, V.addr(inb), V.addr(outb))   # codec.c:620, BASELINE.md
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), 0) == V.ICERR_OK
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), V.addr(outb)) == V.ICERR_OK
    for four in (b"IYUV", b"YV12"):
        assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(V.bmi(64, 48, four)), 0) == V.ICERR_OK
    odd = V.bmi(63, 48, b"I420")
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(odd), 0) == V.ICERR_BADFORMAT          # even dimensions only (codec.c:600,639)
    assert D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(odd), V.addr(outb)) == V.ICERR_BADFORMAT
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(V.bmi(64, 48, b"ZZZZ")), 0) == V.ICERR_BADFORMAT
    wrong = V.bmi(1280, 720, b"H264")
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), V.addr(wrong)) == V.ICERR_BADFORMAT   # size mismatch
    outb.bmiHeader.biCompression = V.fourcc(b"DIVX")
    assert D(cid, None, V.ICM_COMPRESS_QUERY, V.addr(inb), V.addr(outb)) == V.ICERR_BADFORMAT   # output fourcc must be one of codec.c:112-121
    D(cid, None, V.DRV_CLOSE, 0, 0)


def test_begin_without_gpu_is_a_sticky_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cid = open_codec()
    inb, outb = V.bmi(64, 48, b"I420"), V.BITMAPINFO()
    D(cid, None, V.ICM_COMPRESS_GET_FORMAT, V.addr(inb), V.addr(outb))
    assert D(cid, None, V.ICM_COMPRESS_BEGIN, V.addr(inb), V.addr(outb)) == V.ICERR_ERROR
    assert b"x264_encoder_open failed" in V.H.x264vfw_shim_log(cid)
    flags = V.DWORD()
    buf = C.create_string_buffer(outb.bmiHeader.biSizeImage)
    frame = C.create_string_buffer(64 * 48 * 3 // 2)
    icc = V.ICCOMPRESS(lpbiOutput=C.pointer(outb.bmiHeader), lpOutput=C.cast(buf, C.c_void_p), lpbiInput=C.pointer(inb.bmiHeader),
                       lpInput=C.cast(frame, C.c_void_p), lpdwFlags=C.pointer(flags))
    assert D(cid, None, V.ICM_COMPRESS, V.addr(icc), 0) == V.ICERR_ERROR
    assert D(cid, None, V.ICM_COMPRESS_END, 0, 0) == V.ICERR_OK
    assert D(cid, None, V.DRV_CLOSE, 0, 0) == 1

"""ctypes view of libx264gpu_host.so — the B1 shell (x264_* API, include/x264.h) — for Python callers (bench.py, tests)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("X264GPU_HOST_LIB"):
    # an explicit library path (the CPU tests point this at the host sources linked against their stand-in device library)
    H = C.CDLL(os.environ["X264GPU_HOST_LIB"])
else:
    from x264vfw_amd import lib as _gpu  # noqa: F401  (loads libx264gpu.so first; single HIP runtime)
    H = C.CDLL(os.path.join(ROOT, "x264vfw_amd", "libx264gpu_host.so"))
_i = C.c_int


class Vui(C.Structure):
    _fields_ = [(n, _i) for n in ("i_sar_height", "i_sar_width", "i_overscan", "i_vidformat", "b_fullrange", "i_colorprim",
                                  "i_transfer", "i_colmatrix", "i_chroma_loc")]


class Analyse(C.Structure):
    _fields_ = [("intra", C.c_uint), ("inter", C.c_uint)] + [(n, _i) for n in (
        "b_transform_8x8", "i_weighted_pred", "b_weighted_bipred", "i_direct_mv_pred", "i_chroma_qp_offset", "i_me_method",
        "i_me_range", "i_mv_range", "i_mv_range_thread", "i_subpel_refine", "b_chroma_me", "b_mixed_references", "i_trellis",
        "b_fast_pskip", "b_dct_decimate", "i_noise_reduction")] + [("f_psy_rd", C.c_float), ("f_psy_trellis", C.c_float),
        ("b_psy", _i), ("i_luma_deadzone", _i * 2), ("b_psnr", _i), ("b_ssim", _i)]


class Rc(C.Structure):
    _fields_ = [("i_rc_method", _i), ("i_qp_constant", _i), ("i_qp_min", _i), ("i_qp_max", _i), ("i_qp_step", _i), ("i_bitrate", _i),
                ("f_rf_constant", C.c_float), ("f_rf_constant_max", C.c_float), ("f_rate_tolerance", C.c_float),
                ("i_vbv_max_bitrate", _i), ("i_vbv_buffer_size", _i), ("f_vbv_buffer_init", C.c_float), ("f_ip_factor", C.c_float),
                ("f_pb_factor", C.c_float), ("i_aq_mode", _i), ("f_aq_strength", C.c_float), ("b_mb_tree", _i), ("i_lookahead", _i),
                ("b_stat_write", _i), ("psz_stat_out", C.c_char_p), ("b_stat_read", _i), ("psz_stat_in", C.c_char_p),
                ("f_qcompress", C.c_float), ("f_qblur", C.c_float), ("f_complexity_blur", C.c_float), ("psz_zones", C.c_char_p)]


LOGFN = C.CFUNCTYPE(None, C.c_void_p, _i, C.c_char_p, C.c_void_p)


class Param(C.Structure):
    _fields_ = [("cpu", C.c_uint), ("i_threads", _i), ("b_sliced_threads", _i), ("b_deterministic", _i), ("i_width", _i), ("i_height", _i), ("i_csp", _i),
                ("i_level_idc", _i), ("i_frame_total", _i), ("i_nal_hrd", _i), ("vui", Vui), ("i_frame_reference", _i),
                ("i_keyint_max", _i), ("i_keyint_min", _i), ("i_scenecut_threshold", _i), ("b_intra_refresh", _i), ("i_bframe", _i),
                ("i_bframe_adaptive", _i), ("i_bframe_bias", _i), ("i_bframe_pyramid", _i), ("b_open_gop", _i), ("b_bluray_compat", _i),
                ("b_deblocking_filter", _i), ("i_deblocking_filter_alphac0", _i), ("i_deblocking_filter_beta", _i), ("b_cabac", _i),
                ("i_cabac_init_idc", _i), ("b_interlaced", _i), ("b_constrained_intra", _i), ("pf_log", C.c_void_p),
                ("p_log_private", C.c_void_p), ("i_log_level", _i), ("analyse", Analyse), ("rc", Rc), ("b_aud", _i),
                ("b_repeat_headers", _i), ("b_annexb", _i), ("i_sps_id", _i), ("b_vfr_input", _i), ("i_fps_num", C.c_uint32),
                ("i_fps_den", C.c_uint32), ("i_timebase_num", C.c_uint32), ("i_timebase_den", C.c_uint32), ("i_frame_packing", _i),
                ("b_stitchable", _i), ("i_slice_max_size", _i), ("i_slice_max_mbs", _i), ("b_fake_interlaced", _i), ("b_pic_struct", _i), ("i_slice_count", _i)]


class Image(C.Structure):
    _fields_ = [("i_csp", _i), ("i_plane", _i), ("i_stride", _i * 4), ("plane", C.c_void_p * 4)]


class Picture(C.Structure):
    _fields_ = [("i_type", _i), ("i_qpplus1", _i), ("b_keyframe", _i), ("i_pts", C.c_int64), ("i_dts", C.c_int64), ("img", Image),
                ("opaque", C.c_void_p)]


class Nal(C.Structure):
    _fields_ = [("i_ref_idc", _i), ("i_type", _i), ("b_long_startcode", _i), ("i_first_mb", _i), ("i_last_mb", _i),
                ("i_payload", _i), ("p_payload", C.POINTER(C.c_uint8)), ("i_padding", _i)]


class Level(C.Structure):
    _fields_ = [(n, _i) for n in ("level_idc", "mbps", "frame_size", "dpb", "bitrate", "cpb", "mv_range")]


def _sig(name, res, args):
    f = getattr(H, name)
    f.restype, f.argtypes = res, args
    return f


_sig("x264_param_default", None, [C.POINTER(Param)])
_sig("x264_param_default_preset", _i, [C.POINTER(Param), C.c_char_p, C.c_char_p])
_sig("x264_param_parse", _i, [C.POINTER(Param), C.c_char_p, C.c_char_p])
_sig("x264_param_apply_fastfirstpass", None, [C.POINTER(Param)])
_sig("x264_param_apply_profile", _i, [C.POINTER(Param), C.c_char_p])
_sig("x264_picture_alloc", _i, [C.POINTER(Picture), _i, _i, _i])
_sig("x264_picture_clean", None, [C.POINTER(Picture)])
_sig("x264_picture_init", None, [C.POINTER(Picture)])
_sig("x264_encoder_open_157", C.c_void_p, [C.POINTER(Param)])
_sig("x264_encoder_parameters", None, [C.c_void_p, C.POINTER(Param)])
_sig("x264_encoder_headers", _i, [C.c_void_p, C.POINTER(C.POINTER(Nal)), C.POINTER(_i)])
_sig("x264_encoder_encode", _i, [C.c_void_p, C.POINTER(C.POINTER(Nal)), C.POINTER(_i), C.POINTER(Picture), C.POINTER(Picture)])
_sig("x264_encoder_delayed_frames", _i, [C.c_void_p])
_sig("x264_encoder_close", None, [C.c_void_p])
_sig("x264host_write_slice", _i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_void_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)])
_sig("x264host_write_headers", _i, [_i, _i, _i, _i, _i, _i, C.c_uint32, C.c_uint32, _i, _i, C.c_void_p, _i])
_sig("x264host_write_slice_cabac", _i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_void_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)])
_sig("x264host_write_picture", _i, [_i] * 15 + [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)])
_sig("x264host_write_headers_cabac", _i, [_i, _i, _i, _i, _i, _i, C.c_uint32, C.c_uint32, _i, _i, _i, C.c_void_p, _i])
_sig("x264host_get_recon", _i, [C.c_void_p, C.c_void_p])
# tests: the DPB model of the host encoder (host/dpb.hpp) and a CABAC slice writer driven by it (streams with B pictures)
_sig("x264host_dpb_new", C.c_void_p, [_i, _i, _i, _i, _i])
_sig("x264host_dpb_plan_w", _i, [C.c_void_p, _i, _i, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_sig("x264host_dpb_set_direct", None, [C.c_void_p, _i, _i])
_sig("x264host_dpb_plan_wc", _i, [C.c_void_p, _i, _i, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_sig("x264host_dpb_free", None, [C.c_void_p])
_sig("x264host_dpb_info", _i, [C.c_void_p, C.POINTER(_i), C.POINTER(_i)])
_sig("x264host_dpb_plan", _i, [C.c_void_p, _i, _i, _i, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p])
_sig("x264host_dpb_commit", None, [C.c_void_p])
_sig("x264host_write_slice_dpb", _i, [C.c_void_p] + [_i] * 10 + [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)])
_sig("x264host_write_picture_dpb", _i, [C.c_void_p] + [_i] * 11 + [C.c_void_p, C.c_void_p, C.c_void_p, _i, C.POINTER(_i)])
_sig("x264host_write_headers_b", _i, [_i] * 6 + [C.c_uint32, C.c_uint32] + [_i] * 8 + [C.c_void_p, _i])
PIC_IDR, PIC_I, PIC_P, PIC_BREF, PIC_B = range(5)
_sig("x264host_last_decision", _i, [C.c_void_p, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_int32)])
_sig("x264host_last_qpm", C.c_float, [C.c_void_p])
_sig("x264host_pictures_in_flight", C.c_int, [C.c_void_p])
_sig("x264host_pass2_plan", _i, [C.c_void_p, C.c_void_p, C.c_void_p, _i])
LEVELS = (Level * 21).in_dll(H, "x264_levels")

X264_CSP_I420, X264_RC_CQP, X264_RC_CRF, X264_RC_ABR = 1, 0, 1, 2
X264_PARAM_BAD_NAME, X264_PARAM_BAD_VALUE = -1, -2


def write_headers(w, h, level=40, log2_max_frame_num=8, pic_init_qp=23, cqo=0, tick=1, scale=50, num_ref=1, t8x8=0, cabac=0):
    buf = np.zeros(256, np.uint8)
    n = H.x264host_write_headers_cabac(w, h, level, log2_max_frame_num, pic_init_qp, cqo, tick, scale, num_ref, t8x8, cabac, buf.ctypes.data, buf.size)
    assert n > 0
    return bytes(buf[:n])


def write_slice(mbw, mbh, slice_type, qp, pic_init_qp, frame_num, log2_max_frame_num, idr, idr_pic_id, disable_deblock, mbs, lv,
                num_ref=1, num_ref_default=1, t8x8=0, cabac=0, slices=1):
    buf = np.zeros(max(1 << 16, mbs.size * 1200), np.uint8)
    sk = _i()
    mbs = np.ascontiguousarray(mbs)
    lv = np.ascontiguousarray(lv)
    if slices > 1 or slices < -1:           # the picture as several slices, one NAL each (N: x264 slice threads, -N: x264 --slices N)
        n = H.x264host_write_picture(mbw, mbh, slice_type, qp, pic_init_qp, frame_num, log2_max_frame_num, idr, idr_pic_id, disable_deblock, num_ref, num_ref_default,
                                     t8x8, cabac, slices, mbs.ctypes.data, lv.ctypes.data, buf.ctypes.data, buf.size, C.byref(sk))
        assert n > 0
        return bytes(buf[:n]), sk.value
    n = (H.x264host_write_slice_cabac if cabac else H.x264host_write_slice)(mbw, mbh, slice_type, qp, pic_init_qp, frame_num, log2_max_frame_num, idr, idr_pic_id,
                               disable_deblock, num_ref, num_ref_default, t8x8, mbs.ctypes.data, lv.ctypes.data, buf.ctypes.data, buf.size,
                               C.byref(sk))
    assert n > 0
    return bytes(buf[:n]), sk.value


# ---- boundary B2: VfW driver shell (include/vfw_shim.h) ----
DWORD, LONG, WORD = C.c_uint32, C.c_int32, C.c_uint16


class BITMAPINFOHEADER(C.Structure):
    _fields_ = [("biSize", DWORD), ("biWidth", LONG), ("biHeight", LONG), ("biPlanes", WORD), ("biBitCount", WORD),
                ("biCompression", DWORD), ("biSizeImage", DWORD), ("biXPelsPerMeter", LONG), ("biYPelsPerMeter", LONG),
                ("biClrUsed", DWORD), ("biClrImportant", DWORD)]


class BITMAPINFO(C.Structure):
    _fields_ = [("bmiHeader", BITMAPINFOHEADER), ("bmiColors", DWORD * 1)]


class ICOPEN(C.Structure):
    _fields_ = [("dwSize", DWORD), ("fccType", DWORD), ("fccHandler", DWORD), ("dwVersion", DWORD), ("dwFlags", DWORD),
                ("dwError", C.c_ssize_t), ("pV1Reserved", C.c_void_p), ("pV2Reserved", C.c_void_p), ("dnDevNode", DWORD)]


class ICINFO(C.Structure):
    _fields_ = [("dwSize", DWORD), ("fccType", DWORD), ("fccHandler", DWORD), ("dwFlags", DWORD), ("dwVersion", DWORD),
                ("dwVersionICM", DWORD), ("szName", C.c_uint16 * 16), ("szDescription", C.c_uint16 * 128), ("szDriver", C.c_uint16 * 128)]


class ICCOMPRESS(C.Structure):
    _fields_ = [("dwFlags", DWORD), ("lpbiOutput", C.POINTER(BITMAPINFOHEADER)), ("lpOutput", C.c_void_p),
                ("lpbiInput", C.POINTER(BITMAPINFOHEADER)), ("lpInput", C.c_void_p), ("lpckid", C.POINTER(DWORD)),
                ("lpdwFlags", C.POINTER(DWORD)), ("lFrameNum", LONG), ("dwFrameSize", DWORD), ("dwQuality", DWORD),
                ("lpbiPrev", C.POINTER(BITMAPINFOHEADER)), ("lpPrev", C.c_void_p)]


class ICCOMPRESSFRAMES(C.Structure):
    _fields_ = [("dwFlags", DWORD), ("lpbiOutput", C.POINTER(BITMAPINFOHEADER)), ("lOutput", C.c_ssize_t),
                ("lpbiInput", C.POINTER(BITMAPINFOHEADER)), ("lInput", C.c_ssize_t), ("lStartFrame", LONG), ("lFrameCount", LONG),
                ("lQuality", LONG), ("lDataRate", LONG), ("lKeyRate", LONG), ("dwRate", DWORD), ("dwScale", DWORD),
                ("dwOverheadPerFrame", DWORD), ("dwReserved2", DWORD), ("GetData", C.c_void_p), ("PutData", C.c_void_p)]


class VfwConfig(C.Structure):
    """Mirror of X264VFW_CONFIG (include/vfw_shim.h): the reference's CONFIG, field for field."""
    _fields_ = ([(n, _i) for n in ("i_format_version", "i_preset", "i_tuning", "i_profile", "i_level", "i_colorspace", "b_fastdecode", "b_zerolatency",
                                   "i_encoding_type", "i_qp", "i_rf_constant", "i_passbitrate", "i_pass", "b_fast1pass", "b_createstats", "b_updatestats")] +
                [("stats", C.c_char * 260)] + [(n, _i) for n in ("i_output_mode", "i_fourcc", "b_vd_hack")] + [("output_file", C.c_char * 260)] +
                [(n, _i) for n in ("i_sar_width", "i_sar_height", "i_log_level", "b_psnr", "b_ssim", "b_no_asm", "b_disable_decoder")] + [("extra_cmdline", C.c_char * 4096)])


_sig("DriverProc", C.c_ssize_t, [C.c_size_t, C.c_void_p, C.c_uint, C.c_ssize_t, C.c_ssize_t])
_sig("x264vfw_shim_log", C.c_char_p, [C.c_size_t])


def fourcc(s):
    return s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24)


DRV_LOAD, DRV_OPEN, DRV_CLOSE, DRV_FREE, DRV_CONFIGURE, DRV_QUERYCONFIGURE, DRV_USER = 1, 3, 4, 6, 7, 8, 0x4000
ICM_GETSTATE, ICM_SETSTATE, ICM_GETINFO = 0x5000, 0x5001, 0x5002
ICM_COMPRESS_GET_FORMAT, ICM_COMPRESS_GET_SIZE, ICM_COMPRESS_QUERY, ICM_COMPRESS_BEGIN, ICM_COMPRESS, ICM_COMPRESS_END = (
    0x4004, 0x4005, 0x4006, 0x4007, 0x4008, 0x4009)
ICM_DECOMPRESS_QUERY, ICM_COMPRESS_FRAMES_INFO = 0x400b, 0x4046
ICERR_OK, ICERR_UNSUPPORTED, ICERR_BADFORMAT, ICERR_BADSIZE, ICERR_ERROR = 0, -1, -2, -7, -100
AVIIF_KEYFRAME = 0x10


def addr(x):
    return C.addressof(x)


def bmi(w, h, four, bits=12):
    b = BITMAPINFO()
    b.bmiHeader.biSize, b.bmiHeader.biWidth, b.bmiHeader.biHeight = C.sizeof(BITMAPINFOHEADER), w, h
    b.bmiHeader.biPlanes, b.bmiHeader.biBitCount, b.bmiHeader.biCompression = 1, bits, fourcc(four)
    return b

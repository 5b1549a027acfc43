"""ctypes binding of libx264gpu.so (include/x264gpu.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# X264GPU_LIB: an instrumented build of the same library (tools/mb_prof.py: -DMB_PROF), never a different implementation
LIB_PATH = os.environ.get("X264GPU_LIB") or os.path.join(_HERE, "libx264gpu.so")


class X264GpuError(RuntimeError):
    pass


def _load():
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  Importing torch first makes
    # the loader bind libx264gpu.so's NEEDED libamdhip64.so.7 to that already-loaded copy; the other order
    # leaves two runtimes in the process and whichever initialises second sees no device.
    try:
        import torch  # noqa: F401  (plumbing only: device buffers, streams)
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise X264GpuError(
            f"{LIB_PATH} not built — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the HIP hot path)")
    return C.CDLL(LIB_PATH)


_lib = _load()
_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t


class MbRecord(C.Structure):
    """Mirror of struct x264gpu_mb (64 bytes)."""
    _fields_ = [("type", C.c_uint8), ("i16_mode", C.c_uint8), ("chroma_mode", C.c_uint8), ("qp", C.c_uint8),
                ("cbp_luma", C.c_uint8), ("cbp_chroma", C.c_uint8), ("partition", C.c_uint8),
                ("ref", C.c_int8 * 4), ("i4_mode", C.c_uint8 * 16),                ("transform8x8", C.c_uint8), ("mv", (C.c_int16 * 2) * 4),
                ("nnz", C.c_uint32), ("cost", C.c_int32), ("aux", C.c_int32 * 3)]


class Config(C.Structure):
    """Mirror of struct x264gpu_config."""
    _fields_ = [(n, C.c_float if n == "aq_strength" else C.c_int) for n in (
        "width", "height", "streams", "refs", "qp_i", "qp_p", "me_range", "subme", "deblock",
        "deblock_alpha", "deblock_beta", "chroma_qp_offset", "deadzone_inter", "deadzone_intra",
        "dct_decimate", "partitions", "dct8x8", "me_method", "chroma_me", "mixed_refs", "aq_mode", "aq_strength", "fast_pskip", "mv_range", "cabac", "rd", "psy", "psy_rd_q8", "slices", "trellis", "slices_plain", "dpb", "weightb")]


class Pic(C.Structure):
    """Mirror of struct x264gpu_pic (picture control of x264gpu_encode_pictures)."""
    class W(C.Structure):
        _fields_ = [("on", C.c_int8), ("denom", C.c_int8), ("scale", C.c_int16), ("offset", C.c_int16)]
    class WC(C.Structure):
        _fields_ = [("on", C.c_int8 * 2), ("denom", C.c_int8), ("pad", C.c_int8), ("scale", C.c_int16 * 2), ("offset", C.c_int16 * 2)]
    _fields_ = [("slice_type", C.c_int), ("qp", C.c_int), ("poc", C.c_int), ("dst", C.c_int), ("keep", C.c_int), ("nref", C.c_int * 2),
                ("slot", (C.c_int8 * 8) * 2), ("wl0", W * 8), ("blind_dupe", C.c_int), ("qpm", C.c_float), ("wc0", WC * 8), ("direct_temporal", C.c_int), ("direct_auto", C.c_int)]


def make_pic(slice_type, qp, poc, dst, keep, l0=(), l1=()):
    p = Pic(slice_type=slice_type, qp=qp, poc=poc, dst=dst, keep=keep, blind_dupe=-1)
    p.nref[0], p.nref[1] = len(l0), len(l1)
    for i, s in enumerate(l0):
        p.slot[0][i] = s
    for i, s in enumerate(l1):
        p.slot[1][i] = s
    return p


MB_LEVELS = 416

_SIGS = {
    "x264gpu_abi_version": (_i, []),
    "x264gpu_device_count": (_i, []),
    "x264gpu_set_device": (_i, [_i]),
    "x264gpu_last_error": (C.c_char_p, []),
    "x264gpu_malloc": (_i, [C.POINTER(_vp), _sz]),
    "x264gpu_free": (_i, [_vp]),
    "x264gpu_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "x264gpu_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "x264gpu_memcpy_d2d": (_i, [_vp, _vp, _sz, _vp]),
    "x264gpu_memset": (_i, [_vp, _i, _sz, _vp]),
    "x264gpu_stream_sync": (_i, [_vp]),
    "x264gpu_stream_create": (_i, [C.POINTER(_vp)]),
    "x264gpu_stream_destroy": (_i, [_vp]),
    "x264gpu_event_create": (_i, [C.POINTER(_vp)]),
    "x264gpu_event_destroy": (_i, [_vp]),
    "x264gpu_event_record": (_i, [_vp, _vp]),
    "x264gpu_event_sync": (_i, [_vp]),
    "x264gpu_stream_wait_event": (_i, [_vp, _vp]),
    "x264gpu_encoder_create_view": (_i, [C.POINTER(_vp), _vp]),
    "x264gpu_pixel_metric": (_i, [_i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_pixel_var": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_pixel_hadamard_ac": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_dctq4x4": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "x264gpu_dctq8x8": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "x264gpu_intra_predict": (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "x264gpu_hpel_filter": (_i, [_vp, _sz, _i, _i, _i, _i, _vp]),
    "x264gpu_lowres": (_i, [_vp, _i, _i, _i, _vp, _sz, _i, _vp]),
    "x264gpu_mc_luma": (_i, [_vp, _sz, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_mc_avg": (_i, [_vp, _vp, _sz, _i, _vp, _vp]),
    "x264gpu_mc_weight": (_i, [_vp, _sz, _i, _i, _i, _vp, _vp]),
    "x264gpu_mc_chroma": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_csp_img_fill": (C.c_long, [_i, _i, _i, C.POINTER(C.c_long), C.POINTER(_i)]),
    "x264gpu_csp_to_i420": (_i, [C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _i, C.POINTER(_vp), C.POINTER(_i), _vp]),
    "x264gpu_csp_to_i420_batch": (_i, [C.POINTER(_vp), C.POINTER(_i), _sz, _i, _i, _i, _i, _i, C.POINTER(_vp), C.POINTER(_i), _sz, _i, _vp]),
    "x264gpu_encoder_direct_scores": (_i, [_vp, _vp]),
    "x264gpu_encoder_create": (_i, [C.POINTER(_vp), C.POINTER(Config)]),
    "x264gpu_encoder_destroy": (None, [_vp]),
    "x264gpu_encoder_mb_count": (_i, [_vp]),
    "x264gpu_encode_frames": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "x264gpu_encoder_get_recon": (_i, [_vp, _i, _vp, _vp]),
    "x264gpu_encode_pictures": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "x264gpu_pack_levels": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "x264gpu_encoder_get_recon_slot": (_i, [_vp, _i, _i, _vp, _vp]),
    "x264gpu_encoder_stage_count": (_i, []),
    "x264gpu_encoder_stage_name": (C.c_char_p, [_i]),
    "x264gpu_encoder_profile_begin": (_i, [_vp, _i]),
    "x264gpu_encoder_set_debug": (_i, [_vp, _vp]),
    "x264gpu_encoder_profile_end": (_i, [_vp, _vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "x264gpu_encoder_set_qp": (_i, [_vp, _i, _i]),
    "x264gpu_encoder_set_mb_qp_offsets": (_i, [_vp, _vp]),
    "x264gpu_encoder_set_stream_qpms": (_i, [_vp, _vp, _vp]),
    "x264gpu_encoder_set_qpm": (_i, [_vp, C.c_float]),
    "x264gpu_encoder_set_stream_qps": (_i, [_vp, _vp]),
    "x264gpu_encoder_set_lowres_mvs": (_i, [_vp, _vp]),
    "x264gpu_encoder_set_lowres_mvs1": (_i, [_vp, _vp]),
    "x264gpu_encoder_cabac_states": (_i, [_vp, _i, _i, _vp]),
    "x264gpu_trellis_blocks": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "x264gpu_cabac_level_walk": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "x264gpu_encoder_deblock_pictures": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "x264gpu_get_device": (_i, [_vp]),
    "x264gpu_lookahead_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "x264gpu_lookahead_destroy": (None, [_vp]),
    "x264gpu_lookahead_frame_cost": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "x264gpu_lookahead_aq_offsets": (_i, [_vp, _vp, C.c_float, _vp, _vp]),
    "x264gpu_lookahead_aq_offsets_mode": (_i, [_vp, _vp, _i, C.c_float, _vp, _vp]),
    "x264gpu_lookahead_mbtree": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _i, C.c_float, _vp, _vp]),
    "x264gpu_slicetype_create": (_i, [C.POINTER(_vp)] + [_i] * 11),
    "x264gpu_slicetype_destroy": (None, [_vp]),
    "x264gpu_slicetype_put_frame": (_i, [_vp, _i, _vp, _vp]),
    "x264gpu_slicetype_frame_cost": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "x264gpu_slicetype_intra_mbs": (_i, [_vp, _i, _i, _i]),
    "x264gpu_slicetype_cost_est": (_i, [_vp, _i, _i, _i, _i]),
    "x264gpu_slicetype_lowres_mvs": (_vp, [_vp, _i, _i, _i]),
    "x264gpu_slicetype_lowres_mv_costs": (_vp, [_vp, _i, _i, _i]),
    "x264gpu_slicetype_intra_costs": (_vp, [_vp, _i]),
    "x264gpu_slicetype_lowres_costs": (_vp, [_vp, _i, _i, _i]),
    "x264gpu_slicetype_frame_cost_w": (_i, [_vp] + [_i] * 9 + [_vp, _vp]),
    "x264gpu_slicetype_pixel_stats": (_i, [_vp, _i, _vp, _vp, _vp]),
    "x264gpu_slicetype_weight_cost": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "x264gpu_slicetype_chroma_stats": (_i, [_vp, _i, _vp, _vp, _vp]),
    "x264gpu_slicetype_weight_cost_chroma": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "x264gpu_slicetype_set_aq": (_i, [_vp, _i, _vp, _vp]),
    "x264gpu_slicetype_set_row_mode": (_i, [_vp, _i]),
    "x264gpu_slicetype_set_bframe_bias": (_i, [_vp, _i]),
    "x264gpu_slicetype_cost_aq": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "x264gpu_slicetype_clear_propagate": (_i, [_vp, _i, _vp]),
    "x264gpu_slicetype_propagate": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
    "x264gpu_slicetype_finish": (_i, [_vp, _i, C.c_float, C.c_float, _vp, _vp]),
    "x264gpu_slicetype_propagate_cost": (_vp, [_vp, _i]),
}

EXPORTS = tuple(_SIGS)
_missing = []
for _name, (_res, _args) in _SIGS.items():
    try:
        _fn = getattr(_lib, _name)
    except AttributeError:
        _missing.append(_name)
        continue
    _fn.restype, _fn.argtypes = _res, _args
if _missing:
    raise X264GpuError(f"libx264gpu.so lacks symbols declared in include/x264gpu.h: {_missing}")


def check(rc, what="x264gpu call"):
    if rc != 0:
        raise X264GpuError(f"{what} failed ({rc}): {_lib.x264gpu_last_error().decode()}")


def __getattr__(name):
    return getattr(_lib, name)

"""ctypes view of oracle/liboracle.so — the CPU checker.  Imported by tests/, smoke() and bench.py's
cpu_baseline leg only; the product (x264vfw_amd/) never touches it."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_so = os.path.join(ROOT, "oracle", "liboracle.so")
if not os.path.exists(_so):
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
L = C.CDLL(_so)

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i = C.c_int


def _sig(name, res, args):
    f = getattr(L, name)
    f.restype, f.argtypes = res, args
    return f


for _n in ("x264o_sad", "x264o_ssd", "x264o_satd", "x264o_sa8d"):
    _sig(_n, _i, [C.c_void_p, _i, C.c_void_p, _i, _i, _i])
_sig("x264o_var", C.c_uint64, [C.c_void_p, _i, _i, _i])
_sig("x264o_hadamard_ac", C.c_uint64, [C.c_void_p, _i, _i, _i])
_sig("x264o_sub4x4_dct", None, [i16p, C.c_void_p, _i, C.c_void_p, _i])
_sig("x264o_add4x4_idct", None, [C.c_void_p, _i, i16p])
_sig("x264o_sub8x8_dct8", None, [i16p, C.c_void_p, _i, C.c_void_p, _i])
_sig("x264o_add8x8_idct8", None, [C.c_void_p, _i, i16p])
_sig("x264o_dct4x4dc", None, [i16p])
_sig("x264o_idct4x4dc", None, [i16p])
_sig("x264o_dct2x2dc", None, [i16p])


class QuantTables(C.Structure):
    _fields_ = [("quant4_mf", (C.c_uint16 * 16) * 52 * 4), ("quant4_bias", (C.c_uint16 * 16) * 52 * 4),
                ("dequant4_mf", (C.c_int32 * 16) * 6), ("quant8_mf", (C.c_uint16 * 64) * 52 * 2),
                ("quant8_bias", (C.c_uint16 * 64) * 52 * 2), ("dequant8_mf", (C.c_int32 * 64) * 6)]


_sig("x264o_quant_init", None, [C.POINTER(QuantTables), _i, _i])
_sig("x264o_quant_4x4", _i, [i16p, C.c_void_p, C.c_void_p])
_sig("x264o_quant_8x8", _i, [i16p, C.c_void_p, C.c_void_p])
_sig("x264o_quant_4x4_dc", _i, [i16p, _i, _i])
_sig("x264o_quant_2x2_dc", _i, [i16p, _i, _i])
_sig("x264o_dequant_4x4", None, [i16p, C.c_void_p, _i])
_sig("x264o_dequant_8x8", None, [i16p, C.c_void_p, _i])
_sig("x264o_dequant_4x4_dc", None, [i16p, C.c_void_p, _i])
_sig("x264o_dequant_2x2_dc", None, [i16p, i16p, C.c_void_p, _i])
_sig("x264o_decimate_score", _i, [i16p, _i])
_sig("x264o_predict_16x16", None, [C.c_void_p, _i, C.c_void_p, _i, _i])
_sig("x264o_predict_8x8c", None, [C.c_void_p, _i, C.c_void_p, _i, _i])
_sig("x264o_predict_4x4", None, [C.c_void_p, _i, C.c_void_p, _i, _i, _i])
_sig("x264o_predict_8x8_filter", None, [C.c_void_p, _i, C.c_void_p, _i])
_sig("x264o_predict_8x8", None, [C.c_void_p, _i, C.c_void_p, _i])
_sig("x264o_frame_filter", None, [C.POINTER(C.c_void_p), _i, _i, _i, _i])
_sig("x264o_frame_init_lowres", None, [C.c_void_p, _i, _i, _i, C.POINTER(C.c_void_p), _i])
_sig("x264o_pixel_avg_weight", None, [C.c_void_p, _i, C.c_void_p, _i, C.c_void_p, _i, _i, _i, _i])
_sig("x264o_mc_weight", None, [C.c_void_p, _i, C.c_void_p, _i, _i, _i, _i, _i, _i])
_sig("x264o_mc_luma", None, [C.c_void_p, _i, C.POINTER(C.c_void_p), _i, _i, _i, _i, _i, _i, _i])
_sig("x264o_mc_chroma", None, [C.c_void_p, C.c_void_p, _i, C.c_void_p, _i, _i, _i, _i, _i, _i, _i])
_sig("x264o_deblock_luma_edge", None, [C.c_void_p, _i, _i, _i, _i, _i, _i, _i])
_sig("x264o_deblock_chroma_edge", None, [C.c_void_p, _i, _i, _i, _i, _i, _i, _i])

_QT = {}


def quant_tables(dz_inter=21, dz_intra=11):
    key = (dz_inter, dz_intra)
    if key not in _QT:
        t = QuantTables()
        L.x264o_quant_init(C.byref(t), dz_inter, dz_intra)
        _QT[key] = t
    return _QT[key]


def ptr(a, off=0):
    """address of element `off` (flat index) of a contiguous numpy array"""
    return a.ctypes.data + off * a.itemsize


def metric(name, a, b):
    """a, b: (n, h, w) uint8 -> int32[n]"""
    f = getattr(L, "x264o_" + name)
    n, h, w = a.shape
    out = np.empty(n, np.int32)
    for i in range(n):
        out[i] = f(ptr(a, i * h * w), w, ptr(b, i * h * w), w, w, h)
    return out


def dctq4x4(enc, pred, qp, lst):
    """enc,pred: (n,4,4) u8 -> coef, levels (raster), recon"""
    t = quant_tables()
    n = enc.shape[0]
    coef = np.zeros((n, 16), np.int16)
    lev = np.zeros((n, 16), np.int16)
    rec = pred.copy()
    mf = C.addressof(t.quant4_mf[lst][qp])
    bias = C.addressof(t.quant4_bias[lst][qp])
    for i in range(n):
        d = np.zeros(16, np.int16)
        L.x264o_sub4x4_dct(d, ptr(enc, i * 16), 4, ptr(pred, i * 16), 4)
        coef[i] = d
        L.x264o_quant_4x4(d, mf, bias)
        lev[i] = d
        L.x264o_dequant_4x4(d, C.addressof(t.dequant4_mf), qp)
        L.x264o_add4x4_idct(ptr(rec, i * 16), 4, d)
    return coef, lev, rec


def make_padded_planes(img, pad):
    """img (h,w) u8 -> planes (4, h+2p, stride) with plane 0 interior = img; returns (planes, stride)"""
    h, w = img.shape
    stride = (w + 2 * pad + 63) // 64 * 64
    planes = np.zeros((4, h + 2 * pad, stride), np.uint8)
    planes[0, pad:pad + h, pad:pad + w] = img
    return planes, stride


def frame_filter(planes, stride, w, h, pad):
    arr = (C.c_void_p * 4)(*[ptr(planes, (k * planes.shape[1] + pad) * stride + pad) for k in range(4)])
    L.x264o_frame_filter(arr, stride, w, h, pad)


def plane_ptrs(planes, stride, pad):
    return (C.c_void_p * 4)(*[ptr(planes, (k * planes.shape[1] + pad) * stride + pad) for k in range(4)])


# ---- frame pipeline (oracle/encoder.c) ----
import sys
sys.path.insert(0, ROOT)
from x264vfw_amd.lib import Config, MbRecord, MB_LEVELS, Pic, make_pic  # noqa: E402,F401  (plain ctypes structs, no GPU needed)

_sig("x264o_encoder_create", C.c_void_p, [C.POINTER(Config)])
_sig("x264o_encoder_destroy", None, [C.c_void_p])
_sig("x264o_encoder_mb_count", _i, [C.c_void_p])
_sig("x264o_encoder_set_qp", None, [C.c_void_p, _i, _i])
_sig("x264o_encoder_set_mb_qp_offsets", None, [C.c_void_p, C.c_void_p])
_sig("x264o_encoder_set_qpm", None, [C.c_void_p, C.c_float])
_sig("x264o_encoder_encode", _i, [C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_void_p])
_sig("x264o_encoder_encode_pic", _i, [C.c_void_p, C.c_void_p, C.POINTER(Pic), C.c_void_p, C.c_void_p])
_sig("x264o_encoder_get_recon", None, [C.c_void_p, C.c_void_p])
_sig("x264o_encoder_ref_plane", C.c_void_p, [C.c_void_p, _i, C.POINTER(_i), C.POINTER(_i)])
_sig("x264o_lambda", _i, [_i])
_sig("x264o_encoder_set_lowres_mvs", None, [C.c_void_p, C.c_void_p])

MB_DTYPE = np.dtype([("type", "u1"), ("i16_mode", "u1"), ("chroma_mode", "u1"), ("qp", "u1"), ("cbp_luma", "u1"),
                     ("cbp_chroma", "u1"), ("partition", "u1"), ("ref", "i1", 4), ("i4_mode", "u1", 16),
                     ("transform8x8", "u1"), ("mv", "<i2", (4, 2)), ("nnz", "<u4"), ("cost", "<i4"), ("aux", "<i4", 3)])
assert MB_DTYPE.itemsize == 64 == C.sizeof(MbRecord)
SLICE_P, SLICE_B, SLICE_I, SLICE_I_NONIDR = 0, 1, 2, 3
MB_B_DIRECT, MB_B_SKIP, MB_B_INTER, MB_B_8x8 = 7, 8, 9, 10
MB_P_L0, MB_P_8x8, MB_P_SKIP = 4, 5, 6


def mb_ref1(mbs):
    """list-1 reference indices of B inter macroblocks (the first four i4_mode bytes)"""
    return mbs["i4_mode"][..., :4].view(np.int8)


def mb_mv1(mbs):
    """list-1 vectors of B inter macroblocks (the cost / aux bytes)"""
    raw = np.concatenate([mbs["cost"][..., None], mbs["aux"]], axis=-1).astype("<i4")
    return raw.view("<i2").reshape(mbs.shape + (4, 2))



def default_config(width, height, streams=1, **kw):
    c = Config(width=width, height=height, streams=streams, refs=1, qp_i=20, qp_p=23, me_range=16, subme=7,
               deblock=1, deblock_alpha=0, deblock_beta=0, chroma_qp_offset=0, deadzone_inter=21,
               deadzone_intra=11, dct_decimate=1, partitions=2, dct8x8=0, me_method=1, chroma_me=0, mixed_refs=0, aq_mode=0, aq_strength=1.0397, fast_pskip=1, mv_range=0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class OracleEncoder:
    def __init__(self, cfg):
        self.cfg = cfg
        self.h = L.x264o_encoder_create(C.byref(cfg))
        self.n = L.x264o_encoder_mb_count(self.h)

    def encode(self, i420, slice_type):
        mbs = np.zeros(self.n, MB_DTYPE)
        lv = np.zeros((self.n, MB_LEVELS), np.int16)
        i420 = np.ascontiguousarray(i420, np.uint8)
        rc = L.x264o_encoder_encode(self.h, ptr(i420), slice_type, ptr(mbs), ptr(lv))
        assert rc == 0
        return mbs, lv

    def direct_scores(self):
        """--direct auto: (temporal, spatial) skip-probe counts of the last B picture coded with pic.direct_auto"""
        out = np.zeros(2, np.int32)
        L.x264o_encoder_direct_scores(self.h, ptr(out))
        return int(out[0]), int(out[1])

    def encode_pic(self, i420, pic):
        """one picture with explicit control (lib.make_pic): B pictures, explicit reference lists"""
        mbs = np.zeros(self.n, MB_DTYPE)
        lv = np.zeros((self.n, MB_LEVELS), np.int16)
        i420 = np.ascontiguousarray(i420, np.uint8)
        rc = L.x264o_encoder_encode_pic(self.h, ptr(i420), C.byref(pic), ptr(mbs), ptr(lv))
        assert rc == 0, rc
        return mbs, lv

    def cabac_states(self):
        """context variables after the last slice coded (CABAC RD sessions)"""
        out = np.zeros(460, np.uint8)
        L.x264o_encoder_cabac_states.argtypes = [C.c_void_p, C.c_void_p]
        L.x264o_encoder_cabac_states(self.h, out.ctypes.data)
        return out

    def set_qp(self, qp_i, qp_p):
        L.x264o_encoder_set_qp(self.h, qp_i, qp_p)

    def set_qpm(self, qpm):
        """the float quantiser (x264 rc->qpm) of encode()'s following pictures; 0 = the integer one"""
        L.x264o_encoder_set_qpm(self.h, qpm)

    def set_mb_qp_offsets(self, off):
        self._off = None if off is None else np.ascontiguousarray(off, np.float32)      # keep alive
        L.x264o_encoder_set_mb_qp_offsets(self.h, None if self._off is None else ptr(self._off))

    def recon(self):
        w, h = self.cfg.width, self.cfg.height
        out = np.zeros(w * h * 3 // 2, np.uint8)
        L.x264o_encoder_get_recon(self.h, ptr(out))
        return out

    def ref_plane(self, k):
        s, r = _i(), _i()
        p = L.x264o_encoder_ref_plane(self.h, k, C.byref(s), C.byref(r))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (r.value, s.value)).copy()

    def close(self):
        if self.h and L is not None:
            L.x264o_encoder_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()


_sig("x264o_optimize_chroma_2x2_dc", _i, [C.c_void_p, _i])


# ---- lookahead frame cost (oracle/lookahead.c) ----
_sig("x264o_lookahead_create", C.c_void_p, [_i, _i, _i, _i])
_sig("x264o_lookahead_destroy", None, [C.c_void_p])
_sig("x264o_lookahead_frame_cost", _i, [C.c_void_p, C.c_void_p, _i, C.c_void_p, C.c_void_p])


class OracleLookahead:
    def __init__(self, w, h, me_range=16, subme=7):
        self.w, self.h = w, h
        self.nb = ((w + 15) // 16) * ((h + 15) // 16)
        self.la = L.x264o_lookahead_create(w, h, me_range, subme)

    def frame_cost(self, i420, reset=False):
        """-> (out[4] = intra cost, P cost, intra blocks, scored blocks; per-block [intra cost, best cost, packed vector, inter])"""
        out = np.zeros(4, np.int32)
        blocks = np.zeros((self.nb, 4), np.int32)
        i420 = np.ascontiguousarray(i420, np.uint8)
        assert L.x264o_lookahead_frame_cost(self.la, ptr(i420), int(reset), ptr(out), ptr(blocks)) == 0
        return out, blocks

    def close(self):
        if self.la and L is not None:
            L.x264o_lookahead_destroy(self.la)
        self.la = None

    def __del__(self):
        self.close()


# ---- the lookahead's frame costs in x264's structure (oracle/slicetype.c) ----
_sig("x264o_slicetype_create", C.c_void_p, [_i] * 10)
_sig("x264o_slicetype_destroy", None, [C.c_void_p])
_sig("x264o_slicetype_put_frame", _i, [C.c_void_p, _i, C.c_void_p])
_sig("x264o_slicetype_frame_cost", _i, [C.c_void_p, _i, _i, _i, _i, _i])
_sig("x264o_slicetype_intra_mbs", _i, [C.c_void_p, _i, _i])
_sig("x264o_slicetype_cost_est", _i, [C.c_void_p, _i, _i, _i])
_sig("x264o_slicetype_mvs", C.c_void_p, [C.c_void_p, _i, _i, _i])
_sig("x264o_slicetype_mv_costs", C.c_void_p, [C.c_void_p, _i, _i, _i])
_sig("x264o_slicetype_intra_costs", C.c_void_p, [C.c_void_p, _i])
_sig("x264o_slicetype_lowres_costs", C.c_void_p, [C.c_void_p, _i, _i, _i])
_sig("x264o_slicetype_set_aq", None, [C.c_void_p, _i, C.c_void_p])
_sig("x264o_slicetype_cost_aq", _i, [C.c_void_p, _i, _i, _i])
_sig("x264o_slicetype_set_bframe_bias", None, [C.c_void_p, _i])
_sig("x264o_slicetype_frame_cost_w", _i, [C.c_void_p] + [_i] * 9)
_sig("x264o_slicetype_pixel_stats", None, [C.c_void_p, _i, C.c_void_p, C.c_void_p])
_sig("x264o_slicetype_weight_cost", C.c_long, [C.c_void_p] + [_i] * 7)
_sig("x264o_slicetype_chroma_stats", None, [C.c_void_p, _i, C.c_void_p, C.c_void_p])
_sig("x264o_slicetype_weight_cost_chroma", C.c_long, [C.c_void_p, _i, C.c_void_p, C.c_void_p] + [_i] * 6)
_sig("x264o_slicetype_clear_propagate", None, [C.c_void_p, _i])
_sig("x264o_slicetype_propagate", _i, [C.c_void_p, _i, _i, _i, _i, _i, _i])
_sig("x264o_slicetype_finish", _i, [C.c_void_p, _i, C.c_float, C.c_float, C.c_void_p])
_sig("x264o_slicetype_propagate_cost", C.c_void_p, [C.c_void_p, _i])


class OracleSlicetype:
    """slicetype_frame_cost(p0, p1, b) over pictures held in numbered slots"""

    def __init__(self, w, h, slots=8, bframes=3, me_method=1, subme=7, me_range=16, weightb=1, mv_range=512, do_edges=0):
        self.w, self.h = w, h
        self.nb = ((w + 15) // 16) * ((h + 15) // 16)
        self.st = L.x264o_slicetype_create(w, h, slots, bframes, me_method, subme, me_range, weightb, mv_range, do_edges)

    def put(self, slot, i420):
        i420 = np.ascontiguousarray(i420, np.uint8)
        assert L.x264o_slicetype_put_frame(self.st, slot, ptr(i420)) == 0

    def cost(self, s0, s1, sb, d0, d1, weight=None):
        if weight:
            return L.x264o_slicetype_frame_cost_w(self.st, s0, s1, sb, d0, d1, 1, *weight)
        return L.x264o_slicetype_frame_cost(self.st, s0, s1, sb, d0, d1)

    def pixel_stats(self, slot, i420):
        out = np.zeros(2, np.uint64)
        i420 = np.ascontiguousarray(i420, np.uint8)
        L.x264o_slicetype_pixel_stats(self.st, slot, ptr(i420), ptr(out))
        return out

    def weight_cost(self, sf, sr, dist, weight=None):
        return L.x264o_slicetype_weight_cost(self.st, sf, sr, dist, 1 if weight else 0, *(weight or (1, 0, 0)))

    def chroma_stats(self, slot, i420):
        out = np.zeros(4, np.uint64)
        i420 = np.ascontiguousarray(i420, np.uint8)
        L.x264o_slicetype_chroma_stats(self.st, slot, ptr(i420), ptr(out))
        return out

    def weight_cost_chroma(self, sf, fenc, ref, dist, plane, weight=None):
        fenc = np.ascontiguousarray(fenc, np.uint8); ref = np.ascontiguousarray(ref, np.uint8)
        return L.x264o_slicetype_weight_cost_chroma(self.st, sf, ptr(fenc), ptr(ref), dist, plane, 1 if weight else 0, *(weight or (1, 0, 0)))

    def intra_mbs(self, slot, d0):
        return L.x264o_slicetype_intra_mbs(self.st, slot, d0)

    def cost_est(self, slot, d0, d1):
        return L.x264o_slicetype_cost_est(self.st, slot, d0, d1)

    def _arr(self, p, dtype, shape):
        n = int(np.prod(shape))
        return np.frombuffer((C.c_char * (n * np.dtype(dtype).itemsize)).from_address(p), dtype=dtype).reshape(shape).copy()

    def mvs(self, slot, lst, dist):
        return self._arr(L.x264o_slicetype_mvs(self.st, slot, lst, dist), np.int16, (self.nb, 2))

    def mv_costs(self, slot, lst, dist):
        return self._arr(L.x264o_slicetype_mv_costs(self.st, slot, lst, dist), np.int32, (self.nb,))

    def intra_costs(self, slot):
        return self._arr(L.x264o_slicetype_intra_costs(self.st, slot), np.int32, (self.nb,))

    def lowres_costs(self, slot, d0, d1):
        return self._arr(L.x264o_slicetype_lowres_costs(self.st, slot, d0, d1), np.uint16, (self.nb,))

    def cost_aq(self, slot, d0, d1):
        return L.x264o_slicetype_cost_aq(self.st, slot, d0, d1)

    def set_aq(self, slot, aq):
        a = None if aq is None else np.ascontiguousarray(aq, np.float32)
        L.x264o_slicetype_set_aq(self.st, slot, None if a is None else ptr(a))

    def clear_propagate(self, slot):
        L.x264o_slicetype_clear_propagate(self.st, slot)

    def propagate(self, s0, s1, sb, d0, d1, referenced):
        assert L.x264o_slicetype_propagate(self.st, s0, s1, sb, d0, d1, int(referenced)) == 0

    def finish(self, slot, strength, weightdelta=0.0):
        out = np.zeros(self.nb, np.float32)
        assert L.x264o_slicetype_finish(self.st, slot, strength, weightdelta, ptr(out)) == 0
        return out

    def propagate_cost(self, slot):
        return np.minimum(self._arr(L.x264o_slicetype_propagate_cost(self.st, slot), np.int32, (self.nb,)), 32767)

    def close(self):
        if self.st and L is not None:
            L.x264o_slicetype_destroy(self.st)
        self.st = None

    def __del__(self):
        self.close()


_sig("x264o_aq_offsets", None, [C.c_void_p, _i, _i, C.c_float, C.c_void_p])
_sig("x264o_aq_offsets_mode", None, [C.c_void_p, _i, _i, _i, C.c_float, C.c_void_p])
_sig("x264o_mbtree", None, [_i, _i, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i, C.c_float, C.c_void_p])


AQ1 = float(np.float32(1.0) * np.float32(1.0397))       # x264_adaptive_quant_frame's mode-1 strength at --aq-strength 1.0
TREE = float(np.float32(5.0) * (np.float32(1.0) - np.float32(0.6)))       # macroblock_tree_finish's strength at --qcomp 0.6


def aq_offsets_mode(i420, w, h, mode, strength=1.0):
    out = np.zeros(((w + 15) // 16) * ((h + 15) // 16), np.float32)
    i420 = np.ascontiguousarray(i420, np.uint8)
    L.x264o_aq_offsets_mode(ptr(i420), w, h, mode, strength, ptr(out))
    return out


def aq_offsets(i420, w, h, strength=AQ1):
    out = np.zeros(((w + 15) // 16) * ((h + 15) // 16), np.float32)
    i420 = np.ascontiguousarray(i420, np.uint8)
    L.x264o_aq_offsets(ptr(i420), w, h, strength, ptr(out))
    return out


def mbtree(bw, bh, infos, aqs, strength=TREE):
    """infos: list of (blocks x 4) int32 arrays of consecutive pictures, [0] = the one about to be coded; aqs: list of float32 or None"""
    n = len(infos)
    infos = [np.ascontiguousarray(a, np.int32) for a in infos]
    ip = (C.c_void_p * n)(*[a.ctypes.data for a in infos])
    ap = None
    if aqs is not None:
        aqs = [np.ascontiguousarray(a, np.float32) for a in aqs]
        ap = (C.c_void_p * n)(*[a.ctypes.data for a in aqs])
    out = np.zeros(bw * bh, np.float32)
    L.x264o_mbtree(bw, bh, ip, ap, n, strength, ptr(out))
    return out


# ---- oracle/_ref: L-SMASH's H.264 header parser from the reference tree (oracle/lsmash_shim.c, built by oracle/Makefile) ----
class LsSps(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("profile_idc", "constraint_set_flags", "level_idc", "sps_id", "chroma_format_idc", "log2_max_frame_num",
                                         "pic_order_cnt_type", "max_num_ref_frames", "frame_mbs_only_flag", "cropped_width", "cropped_height",
                                         "sar_width", "sar_height", "video_full_range_flag", "colour_primaries", "transfer_characteristics",
                                         "matrix_coefficients", "fixed_frame_rate_flag")] + [("num_units_in_tick", C.c_uint32), ("time_scale", C.c_uint32)]


class LsPps(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("pps_id", "sps_id", "entropy_coding_mode_flag", "num_ref_idx_l0_default_active_minus1", "weighted_pred_flag",
                                         "weighted_bipred_idc", "deblocking_filter_control_present_flag", "redundant_pic_cnt_present_flag")]


class LsSlice(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nal_unit_type", "nal_ref_idc", "slice_type", "idr", "pps_id", "frame_num", "idr_pic_id")]


LSMASH_REF = os.path.join(ROOT, "oracle", "_ref", "liblsmash_ref.so")


def lsmash_parse(stream, max_slices=64):
    """Annex-B bytes -> (sps, pps, [slice headers]) as read by L-SMASH (raises if oracle/_ref was never built)"""
    lib = C.CDLL(LSMASH_REF)
    lib.x264o_lsmash_parse_annexb.restype = _i
    lib.x264o_lsmash_parse_annexb.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(LsSps), C.POINTER(LsPps), C.POINTER(LsSlice), _i, C.POINTER(_i)]
    sps, pps, sl, n = LsSps(), LsPps(), (LsSlice * max_slices)(), _i()
    rc = lib.x264o_lsmash_parse_annexb(bytes(stream), len(stream), C.byref(sps), C.byref(pps), sl, max_slices, C.byref(n))
    assert rc >= 0, f"L-SMASH refused the stream: error {rc}"
    return sps, pps, [sl[i] for i in range(n.value)]


class LsMp4Info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("movie_timescale", "media_timescale", "n_samples", "width", "height", "par_h", "par_v", "n_edits", "avcc_size")] + \
               [(n, C.c_uint64) for n in ("movie_duration", "media_duration", "track_duration", "edit_duration")] + [("edit_start_time", C.c_int64)] + \
               [(n, C.c_int32) for n in ("edit_rate", "primaries", "transfer", "matrix", "full_range")] + \
               [("display_width", C.c_uint32), ("display_height", C.c_uint32), ("avcc", C.c_uint8 * 512)]


class LsMp4Sample(C.Structure):
    _fields_ = [("dts", C.c_uint64), ("cts", C.c_uint64), ("pos", C.c_uint64), ("length", C.c_uint32), ("sync", C.c_uint32)]


def lsmash_read_mp4(path, max_samples=256):
    """mp4 file -> (info, [samples], [sample bytes]) as demuxed by the reference tree's L-SMASH (oracle/_ref)"""
    lib = C.CDLL(LSMASH_REF)
    lib.x264o_lsmash_read_mp4.restype = _i
    lib.x264o_lsmash_read_mp4.argtypes = [C.c_char_p, C.POINTER(LsMp4Info), C.POINTER(LsMp4Sample), _i, C.c_char_p, C.c_size_t]
    info, sm = LsMp4Info(), (LsMp4Sample * max_samples)()
    cap = os.path.getsize(path)
    buf = C.create_string_buffer(cap)
    n = lib.x264o_lsmash_read_mp4(str(path).encode(), C.byref(info), sm, max_samples, buf, cap)
    assert n >= 0, f"L-SMASH refused the file: error {n}"
    out, o = [], 0
    for i in range(min(n, max_samples)):
        out.append(buf.raw[o:o + sm[i].length]); o += sm[i].length
    return info, [sm[i] for i in range(min(n, max_samples))], out


# ---- bitstream checker (oracle/h264dec.cpp) ----
_sig("x264o_h264_decode", _i, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(_i), C.POINTER(_i)])


_sig("x264o_encoder_direct_scores", None, [C.c_void_p, C.c_void_p])
_sig("x264o_h264_last_pocs", _i, [C.c_void_p, _i])
_sig("x264o_h264_last_weighted", None, [C.c_void_p])


def h264_last_weighted():
    """(P slices with an explicit luma weight, P slices with chroma weights) of the last h264_decode"""
    out = np.zeros(2, np.int32)
    L.x264o_h264_last_weighted(ptr(out))
    return int(out[0]), int(out[1])


def h264_last_pocs():
    """picture order counts of the pictures the last h264_decode returned (decoding order)"""
    out = np.zeros(4096, np.int32)
    n = L.x264o_h264_last_pocs(ptr(out), out.size)
    return out[:n].tolist()


def h264_decode(stream, max_frames, w, h):
    """Annex-B bytes -> list of I420 frames (or raises on a syntax error / unsupported feature)"""
    data = np.frombuffer(bytes(stream), np.uint8)
    out = np.zeros(max_frames * w * h * 3 // 2, np.uint8)
    ww, hh = _i(), _i()
    n = L.x264o_h264_decode(ptr(data), len(data), ptr(out), out.nbytes, C.byref(ww), C.byref(hh))
    if n < 0:
        raise ValueError("bitstream rejected by the checker decoder")
    assert (ww.value, hh.value) == (w, h), (ww.value, hh.value)
    fs = w * h * 3 // 2
    return [out[i * fs:(i + 1) * fs].copy() for i in range(min(n, max_frames))]


def dctq8x8(enc, pred, qp, lst):
    """enc,pred: (n,8,8) u8 -> coef, levels (raster), recon — the 8x8 transform path"""
    t = quant_tables()
    n = enc.shape[0]
    coef = np.zeros((n, 64), np.int16)
    lev = np.zeros((n, 64), np.int16)
    rec = pred.copy()
    for i in range(n):
        d = np.zeros(64, np.int16)
        L.x264o_sub8x8_dct8(d, ptr(enc, i * 64), 8, ptr(pred, i * 64), 8)
        coef[i] = d
        L.x264o_quant_8x8(d, C.addressof(t.quant8_mf[lst][qp]), C.addressof(t.quant8_bias[lst][qp]))
        lev[i] = d
        L.x264o_dequant_8x8(d, C.addressof(t.dequant8_mf), qp)
        L.x264o_add8x8_idct8(ptr(rec, i * 64), 8, d)
    return coef, lev, rec


# ---- input colourspace conversion (oracle/csp.c <- /root/reference/csp.c) ----
CSP = dict(I420=1, YV12=2, YV16=3, YV24=4, NV12=5, YUYV=6, UYVY=7, BGR=8, BGRA=9, VFLIP=0x1000)
_sig("x264o_csp_img_fill", C.c_long, [_i, _i, _i, C.POINTER(C.c_long), C.POINTER(_i)])
_sig("x264o_csp_to_i420", _i, [C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(C.c_void_p), C.POINTER(_i), _i, _i, _i, _i, _i])
_sig("x264o_csp_rgb_coefs", None, [_i, _i, C.POINTER(C.c_uint32)])


def csp_img_fill(csp, w, h):
    off, st = (C.c_long * 3)(), (_i * 3)()
    n = L.x264o_csp_img_fill(csp, w, h, off, st)
    return n, list(off), list(st)


def csp_to_i420(buf, csp, w, h, colmatrix709=0, fullrange=0):
    """buf: one contiguous frame laid out as x264vfw_img_fill does -> tight I420 (w*h*3/2 bytes)"""
    buf = np.ascontiguousarray(buf, np.uint8)
    n, off, st = csp_img_fill(csp, w, h)
    assert n == buf.size, (n, buf.size)
    out = np.zeros(w * h * 3 // 2, np.uint8)
    src = (C.c_void_p * 3)(*[buf.ctypes.data + o for o in off])
    dst = (C.c_void_p * 3)(out.ctypes.data, out.ctypes.data + w * h, out.ctypes.data + w * h + (w // 2) * (h // 2))
    rc = L.x264o_csp_to_i420(dst, (_i * 3)(w, w // 2, w // 2), src, (_i * 3)(*st), csp, w, h, colmatrix709, fullrange)
    assert rc == 0
    return out

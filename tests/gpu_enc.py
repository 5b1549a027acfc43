"""Python mirror of the Tier-2 C ABI (include/x264gpu.h x264gpu_encoder_*): thin, no logic of its own."""
import ctypes as C

import numpy as np

import oracle_lib as O
from x264vfw_amd import lib
from x264vfw_amd.lib import MB_LEVELS


class GpuEncoder:
    def __init__(self, cfg):
        import torch
        self.torch = torch
        self.cfg = cfg
        self.h = C.c_void_p()
        lib.check(lib.x264gpu_encoder_create(C.byref(self.h), C.byref(cfg)), "encoder_create")
        self.n = lib.x264gpu_encoder_mb_count(self.h)
        self.S = cfg.streams
        self.d_mb = torch.zeros((self.S, self.n, 64), dtype=torch.uint8, device="cuda")
        self.d_lv = torch.zeros((self.S, self.n, MB_LEVELS), dtype=torch.int16, device="cuda")

    def encode(self, frames, slice_type):
        """frames: list (one per stream) of I420 uint8 arrays -> (mb records [S,n], levels [S,n,416])"""
        t = self.torch
        d_in = t.from_numpy(np.stack(frames)).cuda()
        lib.check(lib.x264gpu_encode_frames(self.h, d_in.data_ptr(), slice_type, self.d_mb.data_ptr(),
                                            self.d_lv.data_ptr(), None), "encode_frames")
        t.cuda.synchronize()
        mb = self.d_mb.cpu().numpy().view(O.MB_DTYPE).reshape(self.S, self.n)
        return mb, self.d_lv.cpu().numpy()

    def encode_pics(self, frames, pics):
        """frames: one I420 array per stream; pics: one lib.Pic per stream (x264gpu_encode_pictures) -> (records [S,n], levels)"""
        t = self.torch
        d_in = t.from_numpy(np.stack(frames)).cuda()
        arr = (lib.Pic * self.S)(*pics)
        lib.check(lib.x264gpu_encode_pictures(self.h, d_in.data_ptr(), arr, self.d_mb.data_ptr(), self.d_lv.data_ptr(), None), "encode_pictures")
        t.cuda.synchronize()
        mb = self.d_mb.cpu().numpy().view(O.MB_DTYPE).reshape(self.S, self.n)
        return mb, self.d_lv.cpu().numpy()

    def direct_scores(self, s=None):
        """--direct auto: (temporal, spatial) skip-probe counts of the last B picture coded with pic.direct_auto (stream s; None: stream 0)"""
        out = np.zeros((self.S, 2), np.int32)
        lib.check(lib.x264gpu_encoder_direct_scores(self.h, out.ctypes.data), "direct_scores")
        return tuple(int(v) for v in out[s or 0])

    def encode_pic(self, frame, pic):
        """single-stream form with the oracle's signature (tests/bgop.py)"""
        assert self.S == 1
        mb, lv = self.encode_pics([frame], [pic])
        return mb[0], lv[0]

    def recon(self, s=0):
        t = self.torch
        w, h = self.cfg.width, self.cfg.height
        out = t.zeros(w * h * 3 // 2, dtype=t.uint8, device="cuda")
        lib.check(lib.x264gpu_encoder_get_recon(self.h, s, out.data_ptr(), None), "get_recon")
        return out.cpu().numpy()

    def cabac_states(self, s=0, sl=0):
        """the 460 context variables the wavefront of (stream, slice) ended the last picture with (CABAC RD sessions)"""
        out = np.zeros(460, np.uint8)
        lib.check(lib.x264gpu_encoder_cabac_states(self.h, s, sl, out.ctypes.data), "cabac_states")
        return out

    def close(self):
        if self.h:
            lib.x264gpu_encoder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuLookahead:
    """Mirror of x264gpu_lookahead_* (include/x264gpu.h)."""

    def __init__(self, w, h, streams=1, me_range=16, subme=7):
        import torch
        self.torch = torch
        self.S = streams
        self.nb = ((w + 15) // 16) * ((h + 15) // 16)
        self.h = C.c_void_p()
        lib.check(lib.x264gpu_lookahead_create(C.byref(self.h), w, h, streams, me_range, subme), "lookahead_create")
        self.d_out = torch.zeros((streams, 4), dtype=torch.int32, device="cuda")
        self.d_blocks = torch.zeros((streams, self.nb, 4), dtype=torch.int32, device="cuda")

    def frame_cost(self, frames, reset=False):
        t = self.torch
        d_in = t.from_numpy(np.stack(frames)).cuda()
        lib.check(lib.x264gpu_lookahead_frame_cost(self.h, d_in.data_ptr(), int(reset), self.d_out.data_ptr(),
                                                   self.d_blocks.data_ptr(), None), "lookahead_frame_cost")
        t.cuda.synchronize()
        return self.d_out.cpu().numpy(), self.d_blocks.cpu().numpy()

    def aq_offsets(self, frames, strength=1.0397):
        t = self.torch
        d_in = t.from_numpy(np.stack(frames)).cuda()
        out = t.zeros((self.S, self.nb), dtype=t.float32, device="cuda")
        lib.check(lib.x264gpu_lookahead_aq_offsets(self.h, d_in.data_ptr(), strength, out.data_ptr(), None), "lookahead_aq_offsets")
        t.cuda.synchronize()
        return out

    def aq_offsets_mode(self, frames, mode, strength=1.0):
        t = self.torch
        d_in = t.from_numpy(np.stack(frames)).cuda()
        out = t.zeros((self.S, self.nb), dtype=t.float32, device="cuda")
        lib.check(lib.x264gpu_lookahead_aq_offsets_mode(self.h, d_in.data_ptr(), mode, strength, out.data_ptr(), None), "lookahead_aq_offsets_mode")
        t.cuda.synchronize()
        return out

    def mbtree(self, d_infos, d_aqs, strength=2.0):
        """d_infos / d_aqs: lists of device tensors of consecutive pictures ([0] = the one about to be coded); d_aqs may be None"""
        t = self.torch
        n = len(d_infos)
        ip = (C.c_void_p * n)(*[x.data_ptr() for x in d_infos])
        ap = (C.c_void_p * n)(*[x.data_ptr() for x in d_aqs]) if d_aqs is not None else None
        out = t.zeros((self.S, self.nb), dtype=t.float32, device="cuda")
        lib.check(lib.x264gpu_lookahead_mbtree(self.h, ip, ap, n, strength, out.data_ptr(), None), "lookahead_mbtree")
        t.cuda.synchronize()
        return out.cpu().numpy()

    def close(self):
        if self.h:
            lib.x264gpu_lookahead_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuSlicetype:
    """Mirror of x264gpu_slicetype_* (include/x264gpu.h): slicetype_frame_cost(p0, p1, b) over pictures held in numbered slots"""

    def __init__(self, w, h, streams=1, slots=8, bframes=3, me_method=1, subme=7, me_range=16, weightb=1, mv_range=512, do_edges=0):
        import torch
        self.torch = torch
        self.S = streams
        self.nb = ((w + 15) // 16) * ((h + 15) // 16)
        self.h = C.c_void_p()
        lib.check(lib.x264gpu_slicetype_create(C.byref(self.h), w, h, streams, slots, bframes, me_method, subme, me_range, weightb, mv_range, do_edges), "slicetype_create")
        self.keep = {}

    def put(self, slot, frames):
        d_in = self.torch.from_numpy(np.stack(frames)).cuda()
        self.keep[slot] = d_in
        lib.check(lib.x264gpu_slicetype_put_frame(self.h, slot, d_in.data_ptr(), None), "slicetype_put_frame")

    def cost(self, s0, s1, sb, d0, d1, weight=None):
        out = np.zeros(self.S, np.int32)
        if weight:
            lib.check(lib.x264gpu_slicetype_frame_cost_w(self.h, s0, s1, sb, d0, d1, 1, *weight, out.ctypes.data, None), "slicetype_frame_cost_w")
        else:
            lib.check(lib.x264gpu_slicetype_frame_cost(self.h, s0, s1, sb, d0, d1, out.ctypes.data, None), "slicetype_frame_cost")
        return out

    def pixel_stats(self, slot):
        out = np.zeros((self.S, 2), np.uint64)
        lib.check(lib.x264gpu_slicetype_pixel_stats(self.h, slot, self.keep[slot].data_ptr(), out.ctypes.data, None), "pixel_stats")
        return out

    def weight_cost(self, sf, sr, dist, weight=None):
        out = np.zeros(self.S, np.int64)
        lib.check(lib.x264gpu_slicetype_weight_cost(self.h, sf, sr, dist, 1 if weight else 0, *(weight or (1, 0, 0)), out.ctypes.data, None), "weight_cost")
        return out

    def chroma_stats(self, slot):
        out = np.zeros((self.S, 4), np.uint64)
        lib.check(lib.x264gpu_slicetype_chroma_stats(self.h, slot, self.keep[slot].data_ptr(), out.ctypes.data, None), "chroma_stats")
        return out

    def weight_cost_chroma(self, sf, sr, dist, plane, weight=None):
        out = np.zeros(self.S, np.int64)
        lib.check(lib.x264gpu_slicetype_weight_cost_chroma(self.h, sf, self.keep[sf].data_ptr(), self.keep[sr].data_ptr(), dist, plane, 1 if weight else 0, *(weight or (1, 0, 0)),
                                                           out.ctypes.data, None), "weight_cost_chroma")
        return out

    def intra_mbs(self, slot, d0, s=0):
        return lib.x264gpu_slicetype_intra_mbs(self.h, slot, d0, s)

    def _dev(self, p, dtype, shape):
        t = self.torch
        assert p, "not computed"
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        buf = np.zeros(n, np.uint8)
        lib.check(lib.x264gpu_memcpy_d2h(buf.ctypes.data, p, n, None), "d2h")
        t.cuda.synchronize()
        return buf.view(dtype).reshape(shape)

    def mvs(self, slot, lst, dist):
        return self._dev(lib.x264gpu_slicetype_lowres_mvs(self.h, slot, lst, dist), np.int16, (self.S, self.nb, 2))

    def mv_costs(self, slot, lst, dist):
        return self._dev(lib.x264gpu_slicetype_lowres_mv_costs(self.h, slot, lst, dist), np.int32, (self.S, self.nb))

    def intra_costs(self, slot):
        return self._dev(lib.x264gpu_slicetype_intra_costs(self.h, slot), np.int32, (self.S, self.nb))

    def lowres_costs(self, slot, d0, d1):
        return self._dev(lib.x264gpu_slicetype_lowres_costs(self.h, slot, d0, d1), np.uint16, (self.S, self.nb))

    def cost_aq(self, slot, d0, d1):
        """i_cost_est_aq of a costed triple, per stream"""
        out = (C.c_int32 * self.S)()
        lib.check(lib.x264gpu_slicetype_cost_aq(self.h, slot, d0, d1, out, None), "cost_aq")
        return list(out)

    def set_aq(self, slot, aq):
        """aq: [blocks] float32 (replicated over the streams) or None"""
        if aq is None:
            lib.check(lib.x264gpu_slicetype_set_aq(self.h, slot, None, None), "set_aq")
            return
        d = self.torch.from_numpy(np.ascontiguousarray(np.tile(np.asarray(aq, np.float32), (self.S, 1)))).cuda()
        lib.check(lib.x264gpu_slicetype_set_aq(self.h, slot, d.data_ptr(), None), "set_aq")
        self.torch.cuda.synchronize()

    def clear_propagate(self, slot):
        lib.check(lib.x264gpu_slicetype_clear_propagate(self.h, slot, None), "clear_propagate")

    def propagate(self, s0, s1, sb, d0, d1, referenced):
        lib.check(lib.x264gpu_slicetype_propagate(self.h, s0, s1, sb, d0, d1, int(referenced), None), "propagate")

    def finish(self, slot, strength, weightdelta=0.0):
        out = self.torch.zeros((self.S, self.nb), dtype=self.torch.float32, device="cuda")
        lib.check(lib.x264gpu_slicetype_finish(self.h, slot, strength, weightdelta, out.data_ptr(), None), "finish")
        self.torch.cuda.synchronize()
        return out.cpu().numpy()

    def propagate_cost(self, slot):
        return np.minimum(self._dev(lib.x264gpu_slicetype_propagate_cost(self.h, slot), np.int32, (self.S, self.nb)), 32767)

    def close(self):
        if self.h:
            lib.x264gpu_slicetype_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

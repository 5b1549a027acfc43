"""Streams with B pictures, host-side helpers for Python callers (bench.py, tests): the coding-order schedule the way x264's lookahead hands
pictures to the encoder (the P or I that closes a run of B pictures first, then under b-pyramid the middle B as a reference, then the others in
display order) and the host library's DPB model (host/dpb.hpp: destination slot, reference lists, frame_num / POC / list modification / marking)
through its exported hooks."""
import ctypes as C

import numpy as np

from x264vfw_amd.lib import Pic


def schedule(types, pyramid=1):
    """display-order picture types ('I' first = IDR, 'i', 'P', 'B') -> coding order list of (display index, host PIC_* type); a run of >= 2 B
    pictures gets its middle one as a B-reference (x264 --b-pyramid normal: index (run - 1) / 2... the lookahead's choice, restated in the host)"""
    PIC = dict(I=0, i=1, P=2, R=3, B=4)
    out, run = [], []
    for i, t in enumerate(types):
        if t == 'B':
            run.append(i)
            continue
        out.append((i, PIC['I'] if (t == 'I') else PIC[t]))
        if len(run) >= 2 and pyramid:
            mid = run[(len(run) - 1) // 2]
            out.append((mid, PIC['R']))
            out += [(j, PIC['B']) for j in run if j != mid]
        else:
            out += [(j, PIC['B']) for j in run]
        run = []
    assert not run, "the last picture of a stream is never a B picture"
    return out


class HostDpb:
    def __init__(self, HL, refs, bframes, pyramid, log2_max_frame_num=4, weightp=0):
        self.HL, self.H = HL, HL.H
        self.weightp = weightp
        self.h = self.H.x264host_dpb_new(refs, bframes, pyramid, log2_max_frame_num, weightp)
        a, b = C.c_int(), C.c_int()
        self.slots = self.H.x264host_dpb_info(self.h, C.byref(a), C.byref(b))
        self.max_dpb, self.num_reorder = a.value, b.value
        self.log2_max_frame_num = log2_max_frame_num
        d = (bframes + 2) * ((1 if pyramid else 0) + 1) * 2
        self.log2_max_poc_lsb = 4
        while (1 << self.log2_max_poc_lsb) <= d * 2:
            self.log2_max_poc_lsb += 1

    def plan(self, ptype, frame, follow=(), weight=None):
        """follow: (coding index, display index) of the non-reference pictures coded right after this one; weight: (scale, denom, offset) of
        reference 0 of a P picture (x264_weights_analyse's result), or None"""
        pic = Pic()
        info = (C.c_int * 8)()
        fc = (C.c_int * max(1, len(follow)))(*[c for c, _ in follow])
        ff = (C.c_int * max(1, len(follow)))(*[f for _, f in follow])
        if weight is not None and len(weight) == 10:          # + (chroma denom, Cb on, Cb scale, Cb offset, Cr on, Cr scale, Cr offset)
            wv = (C.c_int * 10)(*weight)
            self.H.x264host_dpb_plan_wc(self.h, ptype, frame, len(follow), fc, ff, wv, C.byref(pic), info)
        elif weight is not None:
            wv = (C.c_int * 3)(*weight)
            self.H.x264host_dpb_plan_w(self.h, ptype, frame, len(follow), fc, ff, wv, C.byref(pic), info)
        else:
            self.H.x264host_dpb_plan(self.h, ptype, frame, len(follow), fc, ff, C.byref(pic), info)
        return pic, list(info)

    def set_direct(self, pic, temporal, auto=0):
        """--direct temporal / auto for the B picture just planned: into the picture control and the slice header"""
        pic.direct_temporal, pic.direct_auto = int(temporal), int(auto)
        self.H.x264host_dpb_set_direct(self.h, int(temporal), int(auto))

    def commit(self):
        self.H.x264host_dpb_commit(self.h)

    def headers(self, w, h, pic_init_qp, cqo, num_ref_default, t8x8, weightb):
        buf = np.zeros(256, np.uint8)
        n = self.H.x264host_write_headers_b(w, h, 40, self.log2_max_frame_num, pic_init_qp, cqo, 1, 50, num_ref_default, t8x8, 1, self.max_dpb,
                                            self.log2_max_poc_lsb, self.num_reorder, 2 if weightb else 0, 1 if self.weightp else 0, buf.ctypes.data, buf.size)
        assert n > 0
        return bytes(buf[:n])

    def slice(self, mbw, mbh, qp, pic_init_qp, idr_pic_id, disable_deblock, num_ref_default, t8x8, mbs, lv, slices=1):
        buf = np.zeros(max(1 << 16, mbs.size * 1200), np.uint8)
        sk = C.c_int()
        mbs = np.ascontiguousarray(mbs); lv = np.ascontiguousarray(lv)
        n = self.H.x264host_write_picture_dpb(self.h, mbw, mbh, qp, pic_init_qp, self.log2_max_frame_num, self.log2_max_poc_lsb, idr_pic_id, disable_deblock,
                                              num_ref_default, t8x8, slices, mbs.ctypes.data, lv.ctypes.data, buf.ctypes.data, buf.size, C.byref(sk))
        assert n > 0
        return bytes(buf[:n])

    def close(self):
        if self.h:
            self.H.x264host_dpb_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def follow_of(order, k):
    """the non-reference pictures coded right after coding position k"""
    out = []
    for j in range(k + 1, len(order)):
        if order[j][1] != 4:
            break
        out.append((j, order[j][0]))
    return out



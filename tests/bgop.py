"""Helpers for streams with B pictures (tests): a coding-order schedule the way x264's lookahead hands pictures to the encoder (the P or I that
closes a run of B pictures first, then under b-pyramid the middle B as a reference, then the others in display order), the host's DPB model
(host/dpb.hpp through its test hooks) and the closed loop oracle / device records -> host CABAC writer -> checker decoder."""
import ctypes as C

import numpy as np

import oracle_lib as O
from x264vfw_amd.lib import Pic


def schedule(types, pyramid=1):
    """display-order picture types ('I' first = IDR, 'i', 'P', 'B') -> coding order list of (display index, host PIC_* type); a run of >= 2 B
    pictures gets its middle one as a B-reference (x264 --b-pyramid normal: index (run - 1) / 2... the lookahead's choice, restated in the host)"""
    from x264vfw_amd import host_api as HL  # noqa: F401
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
    def __init__(self, HL, refs, bframes, pyramid, log2_max_frame_num=4):
        self.HL, self.H = HL, HL.H
        self.h = self.H.x264host_dpb_new(refs, bframes, pyramid, log2_max_frame_num)
        a, b = C.c_int(), C.c_int()
        self.slots = self.H.x264host_dpb_info(self.h, C.byref(a), C.byref(b))
        self.max_dpb, self.num_reorder = a.value, b.value
        self.log2_max_frame_num = log2_max_frame_num
        d = (bframes + 2) * ((1 if pyramid else 0) + 1) * 2
        self.log2_max_poc_lsb = 4
        while (1 << self.log2_max_poc_lsb) <= d * 2:
            self.log2_max_poc_lsb += 1

    def plan(self, ptype, frame, follow=()):
        """follow: (coding index, display index) of the non-reference pictures coded right after this one"""
        pic = Pic()
        info = (C.c_int * 8)()
        fc = (C.c_int * max(1, len(follow)))(*[c for c, _ in follow])
        ff = (C.c_int * max(1, len(follow)))(*[f for _, f in follow])
        self.H.x264host_dpb_plan(self.h, ptype, frame, len(follow), fc, ff, C.byref(pic), info)
        return pic, list(info)

    def commit(self):
        self.H.x264host_dpb_commit(self.h)

    def headers(self, w, h, pic_init_qp, cqo, num_ref_default, t8x8, weightb):
        buf = np.zeros(256, np.uint8)
        n = self.H.x264host_write_headers_b(w, h, 40, self.log2_max_frame_num, pic_init_qp, cqo, 1, 50, num_ref_default, t8x8, 1, self.max_dpb,
                                            self.log2_max_poc_lsb, self.num_reorder, 2 if weightb else 0, buf.ctypes.data, buf.size)
        assert n > 0
        return bytes(buf[:n])

    def slice(self, mbw, mbh, qp, pic_init_qp, idr_pic_id, disable_deblock, num_ref_default, t8x8, mbs, lv):
        buf = np.zeros(max(1 << 16, mbs.size * 1200), np.uint8)
        sk = C.c_int()
        mbs = np.ascontiguousarray(mbs); lv = np.ascontiguousarray(lv)
        n = self.H.x264host_write_slice_dpb(self.h, mbw, mbh, qp, pic_init_qp, self.log2_max_frame_num, self.log2_max_poc_lsb, idr_pic_id, disable_deblock,
                                            num_ref_default, t8x8, mbs.ctypes.data, lv.ctypes.data, buf.ctypes.data, buf.size, C.byref(sk))
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


def encode_gop(HL, enc, frames, types, cfg, qp_i, qp_p, qp_b, refs, bframes=3, pyramid=1):
    """encodes `frames` (display order) with picture types `types` through `enc` (anything with encode_pic(i420, pic) -> mbs, lv and recon());
    returns (annex-B stream, [recon per coding position], coding order, pocs)"""
    w, h = cfg.width, cfg.height
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    dpb = HostDpb(HL, refs, bframes, pyramid)
    stream = dpb.headers(w, h, qp_p, cfg.chroma_qp_offset, refs, cfg.dct8x8, cfg.weightb)
    order = schedule(types, pyramid)
    recons, pocs = [], []
    idr_id = 0
    for k, (disp, pt) in enumerate(order):
        pic, info = dpb.plan(pt, disp, follow_of(order, k))
        pic.qp = qp_i if pt <= 1 else qp_p if pt == 2 else qp_b if pt == 4 else (qp_p + qp_b) // 2
        mbs, lv = enc.encode_pic(frames[disp], pic)
        stream += dpb.slice(mbw, mbh, pic.qp, qp_p, idr_id, 0 if cfg.deblock else 1, refs, cfg.dct8x8, mbs, lv)
        recons.append(enc.recon())
        pocs.append(pic.poc)
        dpb.commit()
        if pt == 0:
            idr_id += 1
    return stream, recons, order, pocs

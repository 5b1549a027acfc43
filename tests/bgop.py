"""Helpers for streams with B pictures (tests): a coding-order schedule the way x264's lookahead hands pictures to the encoder (the P or I that
closes a run of B pictures first, then under b-pyramid the middle B as a reference, then the others in display order), the host's DPB model
(host/dpb.hpp through its test hooks) and the closed loop oracle / device records -> host CABAC writer -> checker decoder."""
import ctypes as C

import numpy as np

import oracle_lib as O
from x264vfw_amd.lib import Pic


from x264vfw_amd.gop import HostDpb, follow_of, schedule  # noqa: E402,F401


def encode_gop(HL, enc, frames, types, cfg, qp_i, qp_p, qp_b, refs, bframes=3, pyramid=1, weightp=0, pics_out=None, weights=None, direct="spatial"):
    """encodes `frames` (display order) with picture types `types` through `enc` (anything with encode_pic(i420, pic) -> mbs, lv and recon());
    returns (annex-B stream, [recon per coding position], coding order, pocs)"""
    w, h = cfg.width, cfg.height
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    dpb = HostDpb(HL, refs, bframes, pyramid, weightp=weightp)
    stream = dpb.headers(w, h, qp_p, cfg.chroma_qp_offset, refs, cfg.dct8x8, cfg.weightb)
    order = schedule(types, pyramid)
    recons, pocs = [], []
    idr_id = 0
    dscore = [0, 0]                  # h->stat.i_direct_score: [0] temporal, [1] spatial
    for k, (disp, pt) in enumerate(order):
        pic, info = dpb.plan(pt, disp, follow_of(order, k), weight=(weights or {}).get(disp) if pt == 2 else None)
        pic.qp = qp_i if pt <= 1 else qp_p if pt == 2 else qp_b if pt == 4 else (qp_p + qp_b) // 2
        if pt >= 3 and direct != "spatial":
            # --direct temporal; --direct auto: x264's running scores decide (slice_header_init), every macroblock probes both modes
            dpb.set_direct(pic, direct == "temporal" or (direct == "auto" and not dscore[1] > dscore[0]), direct == "auto")
        mbs, lv = enc.encode_pic(frames[disp], pic)
        if pt >= 3 and direct == "auto":
            fs = enc.direct_scores()
            if dscore[0] + dscore[1] > mbw * mbh:
                dscore = [dscore[0] * 9 // 10, dscore[1] * 9 // 10]
            dscore = [dscore[0] + fs[0], dscore[1] + fs[1]]
            if pics_out is not None:
                pics_out.append(("direct", int(pic.direct_temporal), fs))
        if pics_out is not None:
            pics_out.append((pic, mbs))
        stream += dpb.slice(mbw, mbh, pic.qp, qp_p, idr_id, 0 if cfg.deblock else 1, refs, cfg.dct8x8, mbs, lv,
                            slices=(-cfg.slices if cfg.slices_plain else cfg.slices) if cfg.slices > 1 else 1)
        recons.append(enc.recon())
        pocs.append(pic.poc)
        dpb.commit()
        if pt == 0:
            idr_id += 1
    return stream, recons, order, pocs

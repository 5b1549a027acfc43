"""oracle/_ref pin: the parameter sets and slice headers this encoder writes are read back by L-SMASH's H.264 parser — third-party
code inside the reference tree (/root/reference/output/L-SMASH, the parser behind the reference's mp4 muxer, output/mp4_lsmash.c).
oracle/Makefile compiles L-SMASH where it lies together with oracle/lsmash_shim.c into oracle/_ref/liblsmash_ref.so.  This pins the
header syntax (SPS incl. cropping / VUI timing, PPS, slice header up to idr_pic_id) against code that is not ours; macroblock data
stays with oracle/h264dec.cpp."""
import os

import numpy as np
import pytest

import host_lib as HL
import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.skipif(not os.path.exists(O.LSMASH_REF), reason="oracle/_ref/liblsmash_ref.so not built (needs /root/reference; make -C oracle)")


@pytest.mark.parametrize("w,h,level,log2fn,refs,t8x8,tick,scale", [
    (176, 144, 11, 4, 1, 0, 1, 50), (208, 120, 21, 8, 3, 1, 1001, 60000), (1920, 1080, 40, 6, 4, 1, 1, 120), (16, 16, 10, 16, 2, 0, 1, 2)])
def test_parameter_sets_read_back(w, h, level, log2fn, refs, t8x8, tick, scale):
    hdr = HL.write_headers(w, h, level, log2fn, 26, -2, tick, scale, refs, t8x8)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    enc = O.OracleEncoder(O.default_config(w, h, refs=refs, dct8x8=t8x8, partitions=7 if t8x8 else 3))
    f = synth_frames(w, h, 1, seed=9)[0]
    mbs, lv = enc.encode(f, 2)
    nal, _ = HL.write_slice(mbw, mbh, 2, 20, 26, 0, log2fn, 1, 3, 0, mbs, lv, num_ref=1, num_ref_default=refs, t8x8=t8x8)
    sps, pps, sl = O.lsmash_parse(hdr + nal)
    assert (sps.profile_idc, sps.level_idc, sps.chroma_format_idc) == (100 if t8x8 else 66, level, 1)
    assert (sps.log2_max_frame_num, sps.pic_order_cnt_type, sps.max_num_ref_frames, sps.frame_mbs_only_flag) == (log2fn, 2, refs, 1)
    assert (sps.cropped_width, sps.cropped_height) == (w, h)                     # frame cropping of the mod-16 padding
    assert (sps.num_units_in_tick, sps.time_scale, sps.fixed_frame_rate_flag) == (tick, scale, 1)
    assert (pps.entropy_coding_mode_flag, pps.num_ref_idx_l0_default_active_minus1) == (0, refs - 1)
    assert (pps.weighted_pred_flag, pps.weighted_bipred_idc, pps.deblocking_filter_control_present_flag, pps.redundant_pic_cnt_present_flag) == (0, 0, 1, 0)
    assert len(sl) == 1 and (sl[0].nal_unit_type, sl[0].nal_ref_idc, sl[0].slice_type, sl[0].idr, sl[0].idr_pic_id) == (5, 3, 2, 1, 3)


def test_slice_headers_read_back():
    """I / P slices of a short sequence: NAL type, nal_ref_idc, slice_type, frame_num (wrapping at MaxFrameNum), idr_pic_id"""
    w, h, log2fn = 176, 144, 4
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    enc = O.OracleEncoder(O.default_config(w, h, refs=2, partitions=3))
    stream, want = HL.write_headers(w, h, 30, log2fn, 23, 0, 1, 50, 2, 0), []
    frames = synth_frames(w, h, 4, seed=2)
    for i in range(19):                                                          # frame_num runs past 16 = MaxFrameNum
        idr = i == 0
        mbs, lv = enc.encode(frames[i % 4], 2 if idr else 0)
        nal, _ = HL.write_slice(mbw, mbh, 2 if idr else 0, 23, 23, i, log2fn, int(idr), 5, 0, mbs, lv, num_ref=max(1, min(i, 2)), num_ref_default=2)
        stream += nal
        want.append((5 if idr else 1, 3 if idr else 2, 2 if idr else 0, int(idr), i % 16, 5 if idr else 0))
    _, _, sl = O.lsmash_parse(stream)
    assert [(s.nal_unit_type, s.nal_ref_idc, s.slice_type, s.idr, s.frame_num, s.idr_pic_id) for s in sl] == want

"""CPU: B pictures end to end on the checker side — oracle/analyse.c's B analysis (spatial direct, both lists' searches, bi-prediction with implicit
weights, B RD decision) -> records -> the product's host CABAC writer + DPB model (host/cabac.cpp, host/dpb.hpp: frame_num, POC type 0, list
modification, MMCO) -> the checker decoder (oracle/h264dec.cpp: list initialisation / modification / marking, B macroblock layer, spatial direct
and weighted prediction written from the standard) must reproduce the encoder's reconstruction of every picture.  The reference's consumers of B
pictures: output/matroska.c:201, codec.c:1822-1826."""
import numpy as np
import pytest

import bgop
import host_lib as HL
import oracle_lib as O
from synth import synth_frames

MEDIUM = dict(refs=3, dpb=4, weightb=1, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256,
              chroma_qp_offset=-2, trellis=63)


def run(w, h, types, seed, bframes=3, pyramid=1, **over):
    kw = dict(MEDIUM, **over)
    frames = synth_frames(w, h, len(types), seed=seed)
    cfg = O.default_config(w, h, **kw)
    enc = O.OracleEncoder(cfg)
    stream, recons, order, pocs = bgop.encode_gop(HL, enc, frames, types, cfg, 20, 23, 25, kw["refs"], bframes, pyramid)
    dec = O.h264_decode(stream, len(order), w, h)
    assert len(dec) == len(order)
    assert O.h264_last_pocs() == pocs
    for k, (d, r) in enumerate(zip(dec, recons)):
        assert np.array_equal(d, r), f"picture {k} (display {order[k][0]}, type {order[k][1]}) decodes differently"
    return stream, order


@pytest.mark.parametrize("w,h,types,seed,over", [
    (176, 144, "IBBBPBBBP", 5, {}),                                               # medium: bframes 3, b-pyramid, ref 3, weightb
    (96, 80, "IBPBBPBBBPP", 2, {}),                                               # runs of 1, 2, 3 B pictures and consecutive P pictures
    (208, 112, "IPBBBPBPBBP", 3, dict(weightb=0, mixed_refs=0)),
    (176, 144, "IBBBPBBBPBBBP", 6, dict(refs=1, dct8x8=0, trellis=0)),            # ref 1: the DPB still holds 4 pictures under b-pyramid
    (128, 96, "IBBPBBP", 7, dict(refs=5, dpb=5, chroma_me=0, psy_rd_q8=0)),
])
def test_b_pictures_decode_to_the_encoders_reconstruction(w, h, types, seed, over):
    run(w, h, types, seed, **over)


def test_without_pyramid_every_b_is_disposable():
    _, order = run(176, 144, "IBBBPBBP", 9, pyramid=0, dpb=3)
    assert all(t != 3 for _, t in order)


def test_schedule_is_x264s_coding_order():
    assert bgop.schedule("IBBBP") == [(0, 0), (4, 2), (2, 3), (1, 4), (3, 4)]
    assert bgop.schedule("IBBPBP") == [(0, 0), (3, 2), (1, 3), (2, 4), (5, 2), (4, 4)]

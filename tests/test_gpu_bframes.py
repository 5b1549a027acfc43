"""GPU: B pictures through the HIP macroblock loop (k_mb.hip.h BS instantiations: spatial direct, both lists' searches, implicit weighted bi-prediction,
x264's B RD decision, bidirectional refinement) — records, levels, reconstruction and CABAC context variables must equal the CPU checker's picture
by picture, and the stream the host writer makes of the device's records must decode to the device's reconstruction."""
import ctypes as C

import numpy as np
import pytest

import bgop
import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.gpu

MEDIUM = dict(refs=3, dpb=4, weightb=1, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256,
              chroma_qp_offset=-2, trellis=63)
USED_CTX = list(range(3, 11)) + list(range(11, 60)) + list(range(60, 70)) + list(range(73, 276)) + list(range(399, 436))


def describe(mbs_g, mbs_o, i):
    f = lambda m: dict(type=int(m["type"][i]), part=int(m["partition"][i]), d8=int(m["i16_mode"][i]), ref=m["ref"][i].tolist(), ref1=O.mb_ref1(m)[i].tolist(),
                       mv=m["mv"][i].tolist(), mv1=O.mb_mv1(m)[i].tolist(), cbp=(int(m["cbp_luma"][i]), int(m["cbp_chroma"][i])), t8=int(m["transform8x8"][i]))
    return f"\n gpu {f(mbs_g)}\n cpu {f(mbs_o)}"


def run(gpu, w, h, types, seed, streams=1, bframes=3, pyramid=1, weightp=0, weights=None, frames=None, qp_frac=None, direct="spatial", **over):
    from gpu_enc import GpuEncoder
    from x264vfw_amd import host_api as HL
    kw = dict(MEDIUM, **over)
    frames = frames if frames is not None else synth_frames(w, h, len(types), seed=seed)
    cfg = O.default_config(w, h, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(O.default_config(w, h, streams=streams, **kw))
    dpb = bgop.HostDpb(HL, kw["refs"], bframes, pyramid, weightp=weightp)
    dupe_used = 0
    stream = dpb.headers(w, h, 23, cfg.chroma_qp_offset, kw["refs"], cfg.dct8x8, cfg.weightb)
    order = bgop.schedule(types, pyramid)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    recons = []
    dscore, modes = [0, 0], []
    for k, (disp, pt) in enumerate(order):
        pic, _ = dpb.plan(pt, disp, bgop.follow_of(order, k), weight=(weights or {}).get(disp) if pt == 2 else None)
        pic.qp = 20 if pt <= 1 else 23 if pt == 2 else 25 if pt == 4 else 24
        if qp_frac:          # a rate-controlled session's float quantiser: qp + frac / 256 (enters the AQ quantisers before the rounding)
            pic.qpm = pic.qp + qp_frac[k % len(qp_frac)] / 256.0          # (exact in a single float)
        if pt >= 3 and direct != "spatial":          # --direct temporal / auto (the running scores pick the mode as x264's slice_header_init does)
            dpb.set_direct(pic, direct == "temporal" or (direct == "auto" and not dscore[1] > dscore[0]), direct == "auto")
        o_mb, o_lv = og.encode_pic(frames[disp], pic)
        g_mb, g_lv = gg.encode_pics([frames[disp]] * streams, [pic] * streams)
        if pt >= 3 and direct == "auto":
            fs = og.direct_scores()
            for s in range(streams):
                assert gg.direct_scores(s) == fs, f"picture {k}: skip-probe counts of the two direct modes differ ({gg.direct_scores(s)} vs {fs})"
            if dscore[0] + dscore[1] > mbw * mbh:
                dscore = [dscore[0] * 9 // 10, dscore[1] * 9 // 10]
            dscore = [dscore[0] + fs[0], dscore[1] + fs[1]]
            modes.append(int(pic.direct_temporal))
        for s in range(streams):
            bad = np.nonzero(g_mb[s].view(np.uint8).reshape(-1, 64) != o_mb.view(np.uint8).reshape(-1, 64))[0]
            assert bad.size == 0, f"picture {k} (display {disp}, type {pt}) stream {s}: record of macroblock {bad[0]} differs" + describe(g_mb[s], o_mb, bad[0])
            assert np.array_equal(g_lv[s], o_lv), f"picture {k}: levels differ (macroblock {np.nonzero((g_lv[s] != o_lv).any(axis=1))[0][0]})"
            assert np.array_equal(gg.recon(s), og.recon()), f"picture {k}: reconstruction differs"
        assert not kw["rd"] or not kw["cabac"] or pt <= 1 or kw["subme"] < 7 or np.array_equal(gg.cabac_states(0, max(kw.get('slices', 1), 1) - 1)[USED_CTX], og.cabac_states()[USED_CTX]), f"picture {k}: CABAC context variables differ"
        stream += dpb.slice(mbw, mbh, pic.qp, 23, 0, 0 if cfg.deblock else 1, kw["refs"], cfg.dct8x8, g_mb[0], g_lv[0],
                            slices=(-cfg.slices if cfg.slices_plain else cfg.slices) if cfg.slices > 1 else 1)
        recons.append(gg.recon(0))
        if pic.blind_dupe > 0:
            dupe_used += int(((g_mb[0]["ref"] == 1) & ((g_mb[0]["type"] == O.MB_P_L0) | (g_mb[0]["type"] == O.MB_P_8x8))[..., None]).sum())
        dpb.commit()
    dec = O.h264_decode(stream, len(order), w, h)
    for k, (d, r) in enumerate(zip(dec, recons)):
        assert np.array_equal(d, r), f"picture {k} of the device's stream decodes differently"
    return dupe_used


@pytest.mark.parametrize("w,h,types,seed,over", [
    (64, 48, "IBP", 1, dict(partitions=6)),                                         # 16x16 types only: direct / skip / L0 / L1 / BI
    (176, 144, "IBBBP", 5, dict(partitions=6)),
    (176, 144, "IBBBPBBBP", 5, {}),                                                 # medium: bframes 3, b-pyramid, ref 3, weightb
    (96, 80, "IBPBBPBBBPP", 2, {}),
    (208, 112, "IPBBBPBPBBP", 3, dict(weightb=0, mixed_refs=0)),
    (176, 144, "IBBBPBBBPBBBP", 6, dict(refs=1, dct8x8=0, trellis=0)),
    (128, 96, "IBBPBBP", 7, dict(refs=5, dpb=5, chroma_me=0, psy_rd_q8=0)),
    (176, 144, "IBBPBP", 8, dict(dct_decimate=0)),                                  # --no-dct-decimate: B slices decimate all the same (x264 b_dct_decimate)
    (176, 144, "IBBBPBBP", 9, dict(me_method=2)),                                   # --me umh in B slices (presets slow and slower)
    (208, 112, "IBBPBP", 10, dict(me_method=2, me_range=24, refs=4, dpb=4)),
    (128, 96, "IBBPBBP", 11, dict(me_method=0)),                                    # dia
    (128, 96, "IBPBBP", 12, dict(me_method=3, me_range=8)),                         # esa
    (176, 144, "IBBBPBP", 13, dict(trellis=63 + 64)),                               # --trellis 2: the search in the analysis' block encodes and every RD candidate of B macroblocks too
    (128, 96, "IBBPBP", 14, dict(trellis=63 + 64, me_method=2)),
    (176, 144, "IBBBPBBP", 20, dict(subme=9, rd=61)),                               # --subme 9, intra sites: intra_rd_refine of the intra macroblocks of B slices
    (176, 144, "IBBPBP", 21, dict(subme=9, rd=61 | 64, trellis=127, me_method=2)),
    (96, 80, "IBPBBP", 22, dict(subme=9, rd=61, qp_i=30)),
    # RD decisions of B slices priced with CAVLC bit counts (x264 --no-cabac at subme 7 and up: k_mb_slice<.., 1, true>)
    (176, 144, "IBBBPBBP", 51, dict(cabac=0, trellis=0)),
    (96, 80, "IBPBBPBBBPP", 52, dict(cabac=0, trellis=0, refs=1, weightb=0, partitions=0xf07)),
    (176, 144, "IBBPBP", 53, dict(cabac=0, trellis=0, me_method=2, dct8x8=0, psy_rd_q8=0, chroma_qp_offset=0)),
    (208, 112, "IBBBPBP", 54, dict(cabac=0, trellis=0, me_method=0, refs=4, dpb=4, slices=3, slices_plain=1)),
    (128, 96, "IBBPBBP", 55, dict(cabac=0, trellis=0, me_method=3, me_range=8, aq_mode=1, aq_strength=1.0397)),
    # --subme 9 in full: the chosen B inter type's vectors on RD cost — x264_me_refine_qpel_rd per list, x264_me_refine_bidir_rd of the bi-predicted parts (k_mb_b_rdrefine.inc)
    (176, 144, "IBBBP", 20, dict(subme=9, rd=3)),                                   # the inter site alone
    (176, 144, "IBBBPBBP", 33, dict(subme=9, rd=63 | 64)),                          # x264's subme 9: every site + deblock-aware RD
    (96, 80, "IBPBBPBBBPP", 34, dict(subme=9, rd=63, partitions=0xf07)),
    (176, 144, "IBBPBP", 35, dict(subme=9, rd=63 | 64, trellis=127, me_method=2)),  # preset slower's analysis: umh, trellis 2
    (128, 96, "IBBPBBP", 36, dict(subme=9, rd=3, refs=2, weightb=0)),
    (208, 112, "IBBBPBP", 37, dict(subme=9, rd=63 | 64, refs=4, dpb=4, qp_i=20, qp_p=22)),
    (176, 144, "IBBBPBBP", 18, dict(subme=9)),                                      # --subme 9 without its refinement sites: chroma-ME in B slices, 4 + 10 sub-pel iterations
    (208, 112, "IBBPBP", 19, dict(subme=9, me_method=2, rd=1 | 64, refs=4, dpb=4)),
    (176, 144, "IBBBPBBP", 16, dict(rd=1 | 64)),                                    # deblock-aware RD (x264 b_deblock_rdo) in B macroblocks: both lists' motion in the boundary strengths
    (176, 144, "IBBPBP", 17, dict(rd=1 | 64, dct8x8=0, trellis=0, me_method=2)),
    (176, 288, "IBBBPBBP", 15, dict(slices=3)),                                     # x264 slice threads: three wavefronts a B picture, each slice with its own neighbours, contexts and statistics
    (176, 144, "IBBPBP", 16, dict(slices=9, slices_plain=1)),                       # --slices 9: the picture-wide intra count behind the fast-intra decision (speculative passes, EncK.sl_stat)
    (96, 160, "IBPBBP", 17, dict(slices=4, slices_plain=1, refs=2)),
    (176, 288, "IBBBP", 18, dict(slices=4, me_method=2)),
    (176, 144, "IBBPBP", 19, dict(partitions=0x707)),                               # --partitions p8x8,i8x8,i4x4 without b8x8: P slices split, B slices stay 16x16
    (176, 144, "IBBP", 20, dict(partitions=0xf06)),                                 # ... and b8x8 without p8x8
])
def test_b_pictures_bitexact_and_decodable(gpu, w, h, types, seed, over):
    run(gpu, w, h, types, seed, **over)


@pytest.mark.parametrize("w,h,types,seed,over", [
    (176, 144, "IPPPPP", 4, {}),                                                    # P only: ref0 + duplicate (+ more references as the DPB fills)
    (176, 144, "IBBBPBBBPBP", 5, {}),                                               # medium with --weightp 2
    (128, 96, "IPPBBPPP", 8, dict(refs=5, dpb=5, mixed_refs=0)),                    # six list entries: the slow candidate path of indices >= 4
    (96, 80, "IPPPP", 2, dict(refs=2, rd=0, trellis=0, subme=5, psy=0, psy_rd_q8=0)),      # no RD: the winner's quarter-pel refinement on the weighted reference
    (208, 112, "IPPPP", 3, dict(me_method=2, subme=6)),                             # umh: full-pel steps on global memory through the weight
    (176, 144, "IPPP", 7, dict(rd=0, trellis=0, subme=4, psy=0, psy_rd_q8=0)),      # subme 4: one refinement iteration on the duplicate
])
def test_weightp_2_blind_duplicate_bitexact_and_decodable(gpu, w, h, types, seed, over):
    """x264 --weightp 2: the duplicate of reference 0 with luma offset -1 (refined from reference 0's vector, searched in full only for 16x8 / 8x16
    halves whose 8x8 blocks both chose it), weighted fetches in the search / refinement / prediction, the loop filter comparing pictures"""
    assert run(gpu, w, h, types, seed, weightp=2, **over) > 0


@pytest.mark.parametrize("streams,w,h,types,over", [(5, 176, 144, "IBBBPBBP", {}), (3, 128, 96, "IBPBBP", dict(me_method=2, trellis=127)), (4, 96, 208, "IBBPBP", dict(slices=3)),
                                                     (4, 176, 144, "IBBBPBP", dict(subme=9, rd=63 | 64)),                      # --subme 9: the refinement coroutines per stream
                                                     (3, 128, 96, "IBBPBP", dict(subme=9, rd=63 | 64, me_method=2, trellis=127, slices=2, slices_plain=1)),
                                                     (4, 176, 144, "IBBBPBP", dict(cabac=0, trellis=0)),                       # RD on CAVLC bit counts in B slices
                                                     (3, 176, 144, "IBBPBP", dict(rd=0, trellis=0, subme=5, psy_rd_q8=0))])    # B analysis without RD
def test_b_pictures_lock_step_streams_with_distinct_content(gpu, streams, w, h, types, over):
    """the lock-step batch as the bench and the cross-session batcher use it: every stream its own content and quantisers, one launch per picture;
    each stream must equal the CPU checker's encode of that stream alone (records, levels, reconstruction)"""
    from gpu_enc import GpuEncoder
    from x264vfw_amd import host_api as HL
    kw = dict(MEDIUM, **over)
    seqs = [synth_frames(w, h, len(types), seed=50 + 7 * s) for s in range(streams)]
    ogs = [O.OracleEncoder(O.default_config(w, h, **kw)) for _ in range(streams)]
    gg = GpuEncoder(O.default_config(w, h, streams=streams, **kw))
    dpb = bgop.HostDpb(HL, kw["refs"], 3, 1, weightp=2)
    order = bgop.schedule(types, 1)
    for k, (disp, pt) in enumerate(order):
        pic, _ = dpb.plan(pt, disp, bgop.follow_of(order, k))
        pics = []
        for s in range(streams):
            q = type(pic)()
            C.memmove(C.byref(q), C.byref(pic), C.sizeof(pic))
            q.qp = (20 if pt <= 1 else 23 if pt == 2 else 25 if pt == 4 else 24) + s
            pics.append(q)
        g_mb, g_lv = gg.encode_pics([seqs[s][disp] for s in range(streams)], pics)
        for s in range(streams):
            o_mb, o_lv = ogs[s].encode_pic(seqs[s][disp], pics[s])
            bad = np.nonzero(g_mb[s].view(np.uint8).reshape(-1, 64) != o_mb.view(np.uint8).reshape(-1, 64))[0]
            assert bad.size == 0, f"picture {k} (display {disp}, type {pt}) stream {s}: record of macroblock {bad[0]} differs" + describe(g_mb[s], o_mb, bad[0])
            assert np.array_equal(g_lv[s], o_lv) and np.array_equal(gg.recon(s), ogs[s].recon()), f"picture {k} stream {s}"
        dpb.commit()


def test_headline_size_b_pictures_and_weightp_bitexact(gpu):
    """1920x1080 (BASELINE.json's headline geometry), the bench's toolset: one mini-GOP I B B B P plus a second P picture (so that a P picture has two
    references and carries the --weightp 2 duplicate) — records, levels, reconstruction and context variables against the CPU checker, and the
    device's stream through the checker decoder"""
    assert run(gpu, 1920, 1080, "IBBBPP", 21, weightp=2) > 0


@pytest.mark.parametrize("w,h,types,seed,direct,over", [
    (176, 144, "IBBBPBBBP", 5, "temporal", {}),                                   # medium's structure: co-located pictures are P and B-ref pictures
    (96, 80, "IBPBBPBBBPP", 2, "temporal", {}),
    (176, 144, "IBBPBBP", 7, "temporal", dict(refs=5, dpb=5)),
    (208, 112, "IBBBPBBP", 3, "temporal", dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),
    (176, 144, "IBBBPBP", 9, "temporal", dict(me_method=2, partitions=0x707)),
    (176, 144, "IBBBPBBP", 13, "temporal", dict(trellis=127, weightb=0)),
    (176, 144, "IBBBPBBBPBBP", 5, "auto", {}),
    (176, 144, "IBBPBBPBBP", 8, "auto", dict(rd=0, trellis=0, subme=4, psy_rd_q8=0)),
    (128, 96, "IBBBPBBBP", 11, "auto", dict(subme=8, rd=63, me_method=1)),
    (208, 104, "IBBBPBP", 903, "auto", dict(refs=4, dpb=4, slices=3, slices_plain=1, rd=0, subme=4, trellis=0, psy_rd_q8=0)),      # --slices N: a repeated slice pass replaces its own probe counts (fuzz seed 903)
    (200, 112, "IPBBBPBBBP", 911, "auto", dict(slices=5, slices_plain=1, subme=8, rd=63 | 64, trellis=127, dct8x8=0, partitions=3)),
    (176, 288, "IBBBPBBP", 14, "auto", dict(slices=3)),
])
def test_temporal_direct_and_direct_auto_bitexact(gpu, w, h, types, seed, direct, over):
    """x264 --direct temporal (mb_predict_mv_direct16x16_temporal; no direct candidates where a co-located block's reference is out of reach) and
    --direct auto (both modes predicted and probed per macroblock, the counts equal the checker's and pick the next picture's mode)"""
    run(gpu, w, h, types, seed, direct=direct, **over)


def test_direct_auto_multistream(gpu):
    run(gpu, 96, 80, "IBBBPBBP", 11, streams=3, direct="auto")


NORD = dict(rd=0, trellis=0, psy_rd_q8=0)


@pytest.mark.parametrize("w,h,types,seed,over", [
    (176, 144, "IBBBPBBP", 21, dict(NORD, subme=5)),
    (176, 144, "IBBPBP", 22, dict(subme=6, trellis=0)),                              # --subme 6: RD in I / P slices, B slices one level down without it
    (128, 96, "IBBPBBP", 23, dict(NORD, subme=2, refs=1, partitions=0x303, mixed_refs=0, weightb=0)),        # veryfast: the probe alone decides B_SKIP
    (176, 144, "IBBBP", 24, dict(NORD, subme=4, refs=2, mixed_refs=0)),              # faster
    (96, 80, "IBPBBP", 25, dict(NORD, subme=1, refs=1, partitions=0x303, mixed_refs=0, weightb=0, dct8x8=0)),      # superfast: SAD decisions
    (176, 144, "IBBP", 26, dict(NORD, subme=3, me_method=2)),
    (208, 112, "IBBPBP", 27, dict(NORD, subme=5, me_method=0, refs=4, dpb=4)),
    (128, 96, "IBBP", 28, dict(NORD, subme=4, me_method=3, me_range=8)),
    (176, 144, "IBBBPBBBP", 29, dict(NORD, subme=5, partitions=0x707)),              # no b8x8
    (176, 288, "IBBBPBBP", 30, dict(NORD, subme=5, slices=3)),
    (176, 144, "IBBBPBBP", 41, dict(NORD, subme=5, cabac=0)),                        # CAVLC (Main profile --no-cabac): the host's B CAVLC writer, checker's CAVLC B parse
    (96, 80, "IBPBBPBBBPP", 42, dict(NORD, subme=4, cabac=0, refs=2)),
    (176, 144, "IBBBPBBP", 31, dict(subme=6)),                                       # preset fast: --subme 6 --trellis 1: the final encode of the B macroblocks searches too (RD 7)
    (176, 144, "IBBPBP", 32, dict(subme=6, trellis=127)),                            # --trellis 2 at subme 6: off in the B slices' analysis (i_mbrd 0), on in their final encode
    (208, 112, "IBBPBP", 33, dict(subme=6, me_method=2, refs=2, dpb=4, mixed_refs=0)),
])
def test_b_pictures_without_rd_bitexact_and_decodable(gpu, w, h, types, seed, over):
    """x264 below --subme 7 analyses B slices without RD (presets superfast .. fast): x264_macroblock_probe_bskip, the fast-skip order of the 16x16
    searches, SATD / SAD decisions, me_refine_qpel of the winner, SA8D against SATD for the transform size (k_mb_b.inc RD == 0)"""
    run(gpu, w, h, types, seed, **over)



def test_config4_size_slow_toolset_bitexact(gpu):
    """3840x2160 (BASELINE.json configs[3]: slow + --me umh + --subme 9) with that toolset: RD refinement in I, P and B slices (cfg.rd 63: vectors per list,
    bi-predicted pairs, intra modes), deblock-aware RD (bit 6), chroma in the B slices' sub-pel costs, --ref 5, --trellis 2, bframes 3 + b-pyramid + weightb,
    --weightp 2's duplicate, --direct auto — one mini-GOP I B P B plus a second P picture against the CPU checker"""
    assert run(gpu, 3840, 2160, "IBPBP", 41, weightp=2, refs=5, dpb=5, me_method=2, subme=9, rd=63 | 64, trellis=127, direct="auto") >= 0


@pytest.mark.parametrize("types,weights,weightp,over", [
    ("IPPPP", {1: (60, 6, 0), 2: (59, 6, 1), 3: (15, 4, -2), 4: (1, 0, -3)}, 2, {}),              # weighted reference 0 + both duplicates (three indices, one picture)
    ("IBBPBBP", {3: (53, 6, 1), 6: (111, 7, 0)}, 2, {}),
    ("IPPP", {1: (60, 6, 0), 2: (59, 6, 1), 3: (29, 5, 2)}, 1, {}),                               # --weightp 1: the weight alone
    ("IPPP", {1: (1, 0, -128), 2: (60, 6, 0)}, 2, dict(refs=1)),
    ("IPPPP", {1: (60, 6, 0), 2: (59, 6, 1), 3: (58, 6, 0), 4: (57, 6, 1)}, 2, dict(refs=5, dpb=5, me_method=2)),      # seven list entries, umh
    # the chroma planes weighted beside luma: (.., chroma denom, Cb on / scale / offset, Cr on / scale / offset) — chroma-ME costs, skip probe, predictions
    ("IPPP", {1: (58, 6, 3, 5, 1, 30, 4, 1, 29, 6), 2: (60, 6, 0, 6, 1, 61, -2, 1, 62, 1), 3: (59, 6, 1, 6, 1, 60, 0, 1, 66, -3)}, 2, {}),
    ("IBPBP", {2: (60, 6, -2, 6, 1, 62, -3, 0, 1, 0), 4: (66, 6, 2, 5, 0, 1, 0, 1, 31, 2)}, 2, {}),
    ("IPP", {1: (60, 6, 1, 4, 1, 15, 0, 1, 17, -2), 2: (1, 0, 4, 6, 1, 60, 1, 1, 61, 0)}, 1, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),
    ("IPPP", {1: (61, 6, 0, 6, 1, 60, 1, 1, 61, 0), 2: (62, 6, -1, 6, 1, 60, 2, 1, 66, -4)}, 2, dict(subme=8, rd=63, me_method=2, trellis=127)),      # RD refinement's part predictions
    ("IPPP", {1: (58, 6, 3, 5, 1, 30, 4, 1, 29, 6), 2: (60, 6, 0, 6, 1, 61, -2, 1, 62, 1)}, 2, dict(me_method=0, refs=2, chroma_me=0)),
])
def test_explicit_luma_weights_bitexact_and_decodable(gpu, types, weights, weightp, over):
    """a fade with the weights x264_weights_analyse would hand the P pictures: weighted reference 0 in the skip probe, every search and every
    prediction, its two duplicates (x264 weighted_reference_duplicate twice), the reference cache and the loop filter seeing one picture behind three indices"""
    from test_bframes_cpu import fade_frames
    run(gpu, 176, 144, types, 3, weightp=weightp, weights=weights, frames=fade_frames(176, 144, len(types), 3), **over)


@pytest.mark.parametrize("types,over", [
    ("IBBP", dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),                                                # B analysis without RD (presets fast and below)
    ("IBBP", dict(rd=0, trellis=0, subme=2, psy_rd_q8=0, cabac=0, refs=1, mixed_refs=0, weightb=0)),      # ... with CAVLC, veryfast's level
    ("IBBP", dict(direct="temporal")),
    ("IBBBP", dict(direct="auto", me_method=2)),
    ("IBP", dict(subme=9, rd=63 | 64)),                                                                    # x264's subme 9 on medium's other tools
    ("IBP", dict(subme=8, rd=63, refs=2, trellis=127)),
    ("IBBP", dict(subme=7, rd=1, aq_mode=1, aq_strength=1.0397)),
    ("IBBP", dict(cabac=0, trellis=0)),                                                                    # B decisions on CAVLC bit counts
])
def test_round4_paths_at_headline_size(gpu, types, over):
    """1920x1080: the paths this round added, one mini-GOP each against the CPU checker — long vectors, the 4 + 10 sub-pel iterations of subme 8 / 9 and
    the per-row structures only show at this size (the 5-sample sub-pel neighbourhood of the B kernels was found by a 2160p case, not by the small ones)"""
    over = dict(over)
    direct = over.pop("direct", "spatial")
    run(gpu, 1920, 1080, types, 77, direct=direct, **over)


def test_explicit_weights_at_headline_size(gpu):
    """1920x1080 fade: luma + chroma weights on the P pictures of a session with B pictures"""
    from test_bframes_cpu import fade_frames
    types, weights = "IBPBP", {2: (60, 6, -2, 6, 1, 62, -3, 0, 1, 0), 4: (66, 6, 2, 5, 0, 1, 0, 1, 31, 2)}
    run(gpu, 1920, 1080, types, 3, weightp=2, weights=weights, frames=fade_frames(1920, 1080, len(types), 3))


def test_b_pictures_multistream(gpu):
    run(gpu, 96, 80, "IBBBPBBP", 11, streams=3)


@pytest.mark.parametrize("w,h,types,seed,over", [
    (128, 96, "IBBPBP", 14, dict(trellis=63 + 64, me_method=2)),                    # umh + trellis 2 + B: the instantiation HISTORY.md §9 kept at -O2 through round 4
    (128, 96, "IBBPBP", 14, dict(trellis=63, me_method=2, subme=9, rd=63 | 64)),    # ... and the refinement one
    (176, 144, "IBBBP", 5, {}),                                                     # the headline instantiation
])
def test_every_simd_doubly_occupied_streams_stay_identical(gpu, w, h, types, seed, over):
    """2304 streams of the SAME content in one launch: more wavefronts than the 1024 SIMDs of an MI355X, so that every SIMD holds two of them and the
    macroblock loops run interleaved — the regime in which round 4 saw CABAC context variables go wrong after a chroma trellis call in an -O3 build
    of the umh B instantiation (an occupancy-dependent result is the signature of a race or a hazard, not of the arithmetic).  Every stream must equal
    the CPU checker, hence every other stream."""
    run(gpu, w, h, types, seed, streams=2304, **over)


def test_config2_size_medium_toolset_bitexact(gpu):
    """1280x720 (BASELINE.json configs[1]: preset medium, bit-exact vs the reference) as medium really is: bframes 3 + b-pyramid + weightb, --weightp 2's
    duplicate, ref 3 + mixed refs, hex, subme 7 with RD on CABAC sizes + psy-rd, trellis 1, 8x8dct, all partitions — two mini-GOPs against the CPU checker,
    the stream decoded back"""
    assert run(gpu, 1280, 720, "IBBBPBBBP", 72, weightp=2) >= 0

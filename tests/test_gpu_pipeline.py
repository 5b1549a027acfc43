"""-m gpu: Tier-2 frame pipeline (x264gpu_encode_frames) vs the CPU oracle encoder, bit-exact:
macroblock records (types, modes, MVs, cbp, nnz), quantised levels and reconstructed pictures."""
import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.gpu

FIELDS = ("aux", "type", "partition", "i16_mode", "chroma_mode", "qp", "cbp_luma", "cbp_chroma", "ref", "i4_mode", "mv",
          "nnz", "cost")


def compare(tag, mbw, g_mb, o_mb, g_lv, o_lv, g_rec, o_rec):
    for f in FIELDS:
        a, b = g_mb[f], o_mb[f]
        if not np.array_equal(a, b):
            bad = np.nonzero((a != b).reshape(len(a), -1).any(1))[0]
            i = int(bad[0])
            pytest.fail(f"{tag}: field {f} differs in {len(bad)} MBs; first MB {i} (x={i % mbw}, y={i // mbw}) "
                        f"gpu={a[i]} oracle={b[i]} | gpu type={g_mb['type'][i]} oracle type={o_mb['type'][i]} "
                        f"gpu mv={g_mb['mv'][i][0]} oracle mv={o_mb['mv'][i][0]} gpu cost={g_mb['cost'][i]} oracle cost={o_mb['cost'][i]}")
    if not np.array_equal(g_lv, o_lv):
        bad = np.nonzero((g_lv != o_lv).any(1))[0]
        i = int(bad[0])
        pos = np.nonzero(g_lv[i] != o_lv[i])[0]
        pytest.fail(f"{tag}: levels differ in {len(bad)} MBs; first MB {i} (x={i % mbw}, y={i // mbw}, type {o_mb['type'][i]}) "
                    f"at idx {pos[:8]} gpu={g_lv[i][pos[:8]]} oracle={o_lv[i][pos[:8]]}")
    if not np.array_equal(g_rec, o_rec):
        pos = np.nonzero(g_rec != o_rec)[0]
        pytest.fail(f"{tag}: recon differs at {len(pos)} bytes, first offsets {pos[:8]}")


@pytest.mark.parametrize("w,h,slices,kw", [
    (352, 288, 18, dict(partitions=3, refs=2)),
    (352, 288, 6, dict(partitions=7, dct8x8=1, refs=2, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, trellis=63)),
    (640, 272, 17, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5, cabac=1)),
    (320, 400, 9, dict(rd=1, subme=6, partitions=3, qp_i=30, qp_p=33)),
])
def test_plain_slices_carry_the_intra_statistics(gpu, w, h, slices, kw):
    """x264 codes the slices of --slices N one after the other and its fast-intra decision reads the picture's intra count so far; the device runs
    the slices side by side on assumed counts and repeats those whose decisions hang on a wrong one.  Content with scene cuts and partial
    changes puts intra macroblocks into P pictures, so that later slices do depend on earlier ones: same records as the serial oracle,
    and the dependence is really there (the slice-threads oracle, which starts every slice's statistics at zero, decides differently)"""
    from gpu_enc import GpuEncoder
    nfr = 6
    frames = synth_frames(w, h, nfr, seed=w + 3 * h, scene_len=2)
    rng = np.random.default_rng(w * h)
    for i in range(1, nfr):                                   # a few fresh patches per picture: intra macroblocks scattered over the slices
        f = frames[i] = frames[i].copy()
        Y = f[:w * h].reshape(h, w)
        for _ in range(6):
            y0, x0 = int(rng.integers(0, h - 32)), int(rng.integers(0, w - 48))
            Y[y0:y0 + 32, x0:x0 + 48] = rng.integers(0, 256, (32, 48), dtype=np.uint8)
    cfg = O.default_config(w, h, slices=slices, slices_plain=1, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    other = O.OracleEncoder(O.default_config(w, h, slices=slices, **kw)) if slices <= (h + 15) // 16 // 4 else None
    differs = False
    for i, f in enumerate(frames):
        st = 2 if i == 0 else 0
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"{w}x{h} {kw} frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
        if other is not None:
            t_mb, _ = other.encode(f, st)
            differs |= not np.array_equal(t_mb["type"], o_mb["type"])
    assert other is None or differs
    og.close(); gg.close()


@pytest.mark.parametrize("w,h,nfr,kw", [
    (64, 48, 4, {}),
    (176, 144, 5, {}),
    (352, 288, 4, {}),
    (176, 144, 3, dict(deblock=0)),
    (176, 144, 3, dict(partitions=0, dct_decimate=0)),
    (176, 144, 3, dict(qp_i=35, qp_p=38)),
    (176, 144, 3, dict(qp_i=10, qp_p=12, subme=2)),
    (208, 120, 3, dict(subme=5)),      # height not a multiple of 16
    (176, 144, 4, dict(partitions=3)),                       # P16x8 / P8x16 / P8x8 search
    (352, 288, 3, dict(partitions=3, qp_i=28, qp_p=31)),
    (208, 120, 3, dict(partitions=1, subme=4, dct_decimate=0)),
    (64, 48, 3, dict(partitions=3, subme=1)),
    (176, 144, 7, dict(refs=3, partitions=3)),               # medium: --ref 3
    (208, 120, 6, dict(refs=2)),
    (96, 80, 7, dict(refs=4, partitions=1, qp_i=30, qp_p=33)),
    (176, 144, 4, dict(subme=9, partitions=3, refs=2)),      # 4 half-pel + 10 quarter-pel steps: 5 px sub-pel neighbourhood
    (352, 288, 3, dict(subme=8, partitions=3, qp_i=30, qp_p=34)),
    (176, 144, 4, dict(dct8x8=1)),                           # adaptive 8x8 transform on inter macroblocks
    (352, 288, 4, dict(dct8x8=1, partitions=3, refs=2, qp_i=26, qp_p=28)),
    (208, 120, 3, dict(dct8x8=1, qp_i=12, qp_p=14, dct_decimate=0)),
    (176, 144, 4, dict(dct8x8=1, partitions=6)),             # Intra_8x8 next to Intra_4x4 / 16x16
    (352, 288, 4, dict(dct8x8=1, partitions=7, refs=3, qp_i=28, qp_p=31)),   # the medium toolset of this round
    (208, 120, 3, dict(dct8x8=1, partitions=4, qp_i=14, qp_p=16)),
    (64, 48, 3, dict(dct8x8=1, partitions=7, qp_i=38, qp_p=40)),
    (352, 288, 4, dict(me_method=0, subme=0, partitions=0, deblock=0)),      # preset ultrafast: me dia, full-pel only, 16x16 only
    (176, 144, 4, dict(chroma_me=1)),                                        # chroma in the sub-pel costs (subme 7)
    (352, 288, 4, dict(chroma_me=1, dct8x8=1, partitions=7, refs=3, qp_i=28, qp_p=31)),   # medium toolset + chroma-ME
    (208, 120, 4, dict(chroma_me=1, partitions=3, subme=5, refs=2, qp_i=18, qp_p=20)),
    (176, 144, 4, dict(chroma_me=1, partitions=3, subme=9, me_method=0)),
    (64, 48, 3, dict(chroma_me=1, subme=4, partitions=3)),                   # below subme 5 the flag is inert
    (176, 144, 6, dict(mixed_refs=1, refs=3, partitions=3)),                 # mixed refs: per-8x8 / per-half references
    (352, 288, 6, dict(mixed_refs=1, refs=3, partitions=7, dct8x8=1, chroma_me=1, qp_i=28, qp_p=31)),   # x264 medium's ME toolset
    (208, 120, 7, dict(mixed_refs=1, refs=4, partitions=1, subme=5, me_method=0)),
    (176, 144, 8, dict(mixed_refs=1, refs=5, partitions=7, dct8x8=1, subme=5, chroma_me=1, qp_i=27, qp_p=30)),   # BASELINE config 4's ref 5
    (96, 80, 8, dict(refs=5, partitions=0, subme=2, me_method=2)),
    (352, 288, 5, dict(mixed_refs=1, refs=2, partitions=3, me_method=2, chroma_me=1)),
    (64, 48, 4, dict(mixed_refs=1, refs=1, partitions=3)),                   # one reference: the flag is inert
    (176, 144, 4, dict(aq_mode=1)),                                          # variance AQ: a quantiser per macroblock
    (352, 288, 5, dict(aq_mode=1, refs=3, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=24, qp_p=27)),
    (208, 120, 4, dict(aq_mode=1, aq_strength=1.55955, partitions=6, dct8x8=1, qp_i=40, qp_p=44)),       # strong AQ near the top of the range
    (96, 80, 4, dict(aq_mode=1, aq_strength=0.51985, partitions=3, qp_i=4, qp_p=6, deblock=0)),          # ... and near the bottom
    (176, 144, 4, dict(me_method=3)),                                        # --me esa: exhaustive search, 16x16 only
    (208, 120, 4, dict(me_method=3, partitions=3, refs=2, me_range=8, chroma_me=1)),      # ... in every partition
    (96, 80, 4, dict(me_method=3, partitions=3, refs=3, mixed_refs=1, me_range=16, subme=5)),
    (176, 144, 4, dict(me_method=2)),                                        # --me umh, 16x16 only
    (352, 288, 4, dict(me_method=2, partitions=3, refs=2, chroma_me=1)),     # umh in every partition
    (352, 288, 3, dict(me_method=2, partitions=7, dct8x8=1, refs=4, subme=9, chroma_me=1, qp_i=26, qp_p=29)),   # preset slow-like (BASELINE config 4 toolset)
    (208, 120, 4, dict(me_method=2, partitions=3, me_range=24, subme=5)),
    (720, 304, 3, dict(me_method=2, partitions=3, me_range=32, qp_i=30, qp_p=33)),   # wide: long cross / many grid rings
    (96, 80, 4, dict(me_method=2, partitions=1, me_range=8, subme=2)),
    (176, 144, 4, dict(me_method=0, partitions=3, subme=5, refs=2)),         # diamond search in every partition
    (208, 120, 3, dict(me_method=0, subme=2, me_range=8)),
    (176, 144, 4, dict(fast_pskip=0, partitions=3, refs=2)),                 # --no-fast-pskip: P_Skip only from empty 16x16 blocks
    (176, 144, 4, dict(fast_pskip=0, subme=1)),
    (352, 288, 4, dict(mv_range=32, partitions=3, refs=2, subme=5)),         # a low level's vector range: limits bite inside the picture
    (208, 120, 4, dict(mv_range=64, me_method=2, me_range=32)),
    (176, 144, 4, dict(subme=3, partitions=3, refs=2)),                      # subme 3 / 4: quarter-pel predictors, fewer refinement steps
    (176, 144, 4, dict(subme=4, partitions=7, dct8x8=1, refs=3, mixed_refs=1)),
    (96, 80, 5, dict(subme=0, partitions=3, refs=3, mixed_refs=1)),          # full-pel only with partitions
    (176, 144, 5, dict(cabac=1, partitions=3, refs=2)),                      # CABAC session: P8x8 costs without the CAVLC sub-type / P_8x8ref0 terms
    (352, 288, 5, dict(cabac=1, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5)),
    (176, 144, 4, dict(slices=2, partitions=3, refs=2)),                     # x264 slice threads: slices analysed on their own, no filtering across them
    (96, 208, 5, dict(slices=3, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5)),
    (64, 272, 4, dict(slices=4, aq_mode=1, partitions=7, dct8x8=1, qp_i=30, qp_p=34)),
    (352, 288, 4, dict(slices=4, me_method=2, partitions=3, refs=2)),
    (48, 336, 4, dict(slices=5, partitions=3, refs=2, qp_i=12, qp_p=15, dct_decimate=0, me_method=3, me_range=8)),
    (176, 144, 4, dict(slices=9, slices_plain=1, partitions=3, refs=2)),     # x264 --slices N: the same split down to one macroblock row, the loop filter crosses the boundaries
    (96, 208, 5, dict(slices=5, slices_plain=1, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, subme=5, cabac=1)),
    (64, 272, 4, dict(slices=17, slices_plain=1, aq_mode=1, partitions=7, dct8x8=1, qp_i=30, qp_p=34)),
    (64, 48, 3, dict(rd=1, subme=6, partitions=0)),                          # RD mode decision (x264 subme 6 / 7, CAVLC bit counts): I16x16 / P16x16 / skip only
    (176, 144, 4, dict(rd=1, subme=6, partitions=2)),                        # + Intra_4x4
    (176, 144, 4, dict(rd=1, subme=6, partitions=1)),                        # + P16x8 / P8x16 / P8x8
    (176, 144, 6, dict(rd=1, subme=7, partitions=3, refs=2, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),      # psy-RD at x264's default strength
    (176, 144, 6, dict(rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),   # preset medium --no-cabac
    (352, 288, 4, dict(rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2)),
    (96, 208, 4, dict(rd=1, subme=6, slices=3, partitions=7, dct8x8=1, refs=2, psy=1, psy_rd_q8=102, chroma_qp_offset=-1, aq_mode=1)),
    (96, 80, 4, dict(rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0)),
    (208, 120, 4, dict(rd=1, subme=6, partitions=7, dct8x8=1, qp_i=44, qp_p=47, psy=1, psy_rd_q8=512, me_method=3, me_range=8)),
])
def test_pipeline_bitexact(gpu, w, h, nfr, kw):
    from gpu_enc import GpuEncoder
    frames = synth_frames(w, h, nfr, seed=w * 7 + h)
    cfg = O.default_config(w, h, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    mbw = (w + 15) // 16
    for i, f in enumerate(frames):
        st = 2 if i in (0, 5) else 0                        # a second IDR empties the DPB again
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"{w}x{h} {kw} frame {i}", mbw, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
    og.close(); gg.close()


CABAC_CTX_I = list(range(3, 11)) + list(range(60, 70)) + list(range(73, 276)) + list(range(399, 436))        # contexts an I slice uses
CABAC_CTX_P = CABAC_CTX_I + list(range(11, 24)) + list(range(40, 60))


@pytest.mark.parametrize("w,h,nfr,kw", [
    (64, 48, 3, dict(partitions=0, subme=6)),
    (176, 144, 4, dict(partitions=2, subme=6)),
    (176, 144, 4, dict(partitions=1, refs=2, subme=6)),
    (176, 144, 4, dict(partitions=3, refs=3, mixed_refs=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, subme=7)),
    (176, 144, 4, dict(partitions=6, dct8x8=1, subme=7)),
    (176, 144, 6, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, subme=7)),      # preset medium (I / P)
    (352, 288, 4, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2, subme=7)),
    (96, 208, 4, dict(slices=3, partitions=7, dct8x8=1, refs=2, psy=1, psy_rd_q8=102, chroma_qp_offset=-1, aq_mode=1, subme=6)),
    (96, 80, 4, dict(partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0, subme=6)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, qp_i=44, qp_p=47, psy=1, psy_rd_q8=512, me_method=3, me_range=8, subme=6)),
    # trellis quantisation of the final encode (x264 --trellis 1 = every site = 63), and site by site: 1 inter 4x4, 2 inter 8x8, 4 chroma,
    # 8 Intra_16x16, 16 Intra_4x4, 32 Intra_8x8
    (176, 144, 4, dict(partitions=1, subme=6, trellis=1)),
    (176, 144, 4, dict(partitions=5, dct8x8=1, subme=7, trellis=2)),
    (176, 144, 4, dict(partitions=1, subme=6, trellis=4)),
    (64, 48, 3, dict(partitions=0, subme=6, trellis=8)),
    (176, 144, 4, dict(partitions=2, subme=6, trellis=16)),
    (176, 144, 4, dict(partitions=4, dct8x8=1, subme=7, trellis=32)),
    (176, 144, 6, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, subme=7, trellis=63)),       # preset medium (I / P)
    (352, 288, 4, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2, subme=7, trellis=63)),
    (96, 208, 4, dict(slices=3, partitions=7, dct8x8=1, refs=2, aq_mode=1, subme=6, trellis=63)),
    (96, 208, 4, dict(slices=13, slices_plain=1, partitions=7, dct8x8=1, refs=2, aq_mode=1, subme=7, psy=1, psy_rd_q8=256, trellis=63)),       # --slices 13 (one row each)
    (96, 80, 4, dict(partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0, subme=6, trellis=63)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, qp_i=44, qp_p=47, subme=6, trellis=63)),
    # --trellis 2 (+ 64): the search also inside the intra analysis' block encodes and in every RD candidate
    (176, 144, 4, dict(partitions=2, subme=6, trellis=127)),
    (176, 144, 5, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, subme=7, trellis=127)),
    (96, 208, 4, dict(slices=3, partitions=7, dct8x8=1, refs=2, aq_mode=1, subme=6, trellis=127)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, qp_i=44, qp_p=47, me_method=2, subme=6, trellis=127)),
])
def test_pipeline_cabac_rd_bitexact(gpu, w, h, nfr, kw):
    """RD mode decision in a CABAC session (x264 subme 6 / 7 at preset medium's entropy coder): the device carries the slice's context
    variables through the macroblock loop and prices every candidate with x264's size-only coder on a copy of them.  Records, levels and
    reconstruction equal the oracle's, and so do the context variables the last slice ends every picture with"""
    from gpu_enc import GpuEncoder
    frames = synth_frames(w, h, nfr, seed=w * 5 + h)
    cfg = O.default_config(w, h, cabac=1, rd=1, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    ns = kw.get("slices", 1)
    for i, f in enumerate(frames):
        st = 2 if i == 0 else 0
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"{w}x{h} {kw} frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
        used = CABAC_CTX_I if st == 2 else CABAC_CTX_P
        np.testing.assert_array_equal(gg.cabac_states(0, ns - 1)[used], og.cabac_states()[used], err_msg=f"context variables after picture {i}")
    og.close(); gg.close()


@pytest.mark.parametrize("w,h,nfr,kw", [
    # RD refinement (x264 --subme 8, i_mbrd 2), site by site (cfg.rd = 1 | sites << 1): 1 the vectors of the P partitions (x264_me_refine_qpel_rd),
    # 2 the Intra_16x16 mode, 4 the chroma mode, 8 the Intra_4x4 modes, 16 the Intra_8x8 modes (intra_rd_refine); 63 = x264's subme 8
    (176, 144, 4, dict(partitions=0, rd=3)),                                                              # 16x16 only: whole-macroblock candidates
    (176, 144, 4, dict(partitions=1, refs=2, rd=3)),                                                       # P8x8 / 16x8 / 8x16 parts, 4x4 transform
    (176, 144, 4, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, rd=3)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, refs=2, qp_i=30, qp_p=34, me_method=2, rd=3, trellis=63)),
    (176, 144, 4, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, rd=3, trellis=127)),       # trellis 2: the search in the part encodes too
    (176, 144, 3, dict(partitions=0, rd=1 | 2 << 1)),                                                      # Intra_16x16 modes
    (176, 144, 3, dict(partitions=6, dct8x8=1, rd=1 | 4 << 1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),   # chroma modes
    (176, 144, 3, dict(partitions=2, rd=1 | 8 << 1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),             # Intra_4x4 modes
    (176, 144, 3, dict(partitions=4, dct8x8=1, rd=1 | 16 << 1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),  # Intra_8x8 modes
    (176, 144, 4, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, rd=63, trellis=63)),      # x264 --subme 8 on medium's toolset
    (208, 120, 4, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2, rd=63, trellis=127)),                  # ... on slow's: umh, trellis 2
    (96, 80, 3, dict(partitions=7, dct8x8=1, qp_i=10, qp_p=12, rd=63)),
    # deblock-aware RD (x264 --subme 9, h->mb.b_deblock_rdo; cfg.rd bit 6): every whole-macroblock candidate is measured after x264_macroblock_deblock
    (176, 144, 4, dict(partitions=7, dct8x8=0, refs=2, rd=1 | 64, qp_i=28, qp_p=30)),                                                            # 4x4 transform: all three internal edges
    (176, 144, 4, dict(partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, rd=1 | 64, trellis=63)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, refs=2, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2, rd=63 | 64, trellis=127)),          # with the refinement: its whole-macroblock candidates too
    (96, 80, 3, dict(partitions=7, dct8x8=1, qp_i=12, qp_p=14, rd=1 | 64)),                                                                        # quantisers at the filter's threshold
    (176, 144, 3, dict(partitions=7, dct8x8=1, rd=63 | 64, psy=1, psy_rd_q8=256, deblock_alpha=-1, deblock_beta=2, qp_i=24, qp_p=27)),
])
def test_pipeline_rd_refinement_bitexact(gpu, w, h, nfr, kw):
    """RD refinement on the device (REF instantiations of the macroblock loop, k_mb_refine.inc) against oracle/analyse.c's restatement of
    x264_me_refine_qpel_rd / intra_rd_refine: records, levels, reconstruction and context variables"""
    from gpu_enc import GpuEncoder
    frames = synth_frames(w, h, nfr, seed=w * 5 + h)
    cfg = O.default_config(w, h, cabac=1, subme=8, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    for i, f in enumerate(frames):
        st = 2 if i == 0 else 0
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"{w}x{h} {kw} frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
        used = CABAC_CTX_I if st == 2 else CABAC_CTX_P
        np.testing.assert_array_equal(gg.cabac_states(0, 0)[used], og.cabac_states()[used], err_msg=f"context variables after picture {i}")
    og.close(); gg.close()


@pytest.mark.parametrize("kw", [dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=2, mixed_refs=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2),
                                dict(cabac=0, rd=1, subme=6, partitions=3, refs=2), dict(cabac=1, rd=1, subme=7, slices=2, partitions=7, dct8x8=1, aq_mode=1)])
def test_pipeline_rd_multistream(gpu, kw):
    """RD sessions with several streams in lock-step: every stream (its own context variables, |mvd| and total_coeff side data) equals its own
    single-stream oracle encode, context variables included"""
    from gpu_enc import GpuEncoder
    w, h, S, nfr = 96, 144, 4, 3
    seqs = [synth_frames(w, h, nfr, seed=300 + 7 * s) for s in range(S)]
    gg = GpuEncoder(O.default_config(w, h, streams=S, **kw))
    ogs = [O.OracleEncoder(O.default_config(w, h, **kw)) for _ in range(S)]
    for i in range(nfr):
        st = 2 if i == 0 else 0
        g_mb, g_lv = gg.encode([seqs[s][i] for s in range(S)], st)
        for s in range(S):
            o_mb, o_lv = ogs[s].encode(seqs[s][i], st)
            compare(f"stream {s} frame {i}", (w + 15) // 16, g_mb[s], o_mb, g_lv[s], o_lv, gg.recon(s), ogs[s].recon())
            if kw["cabac"]:
                used = CABAC_CTX_I if st == 2 else CABAC_CTX_P
                np.testing.assert_array_equal(gg.cabac_states(s, kw.get("slices", 1) - 1)[used], ogs[s].cabac_states()[used], err_msg=f"stream {s} picture {i}")
    for o in ogs:
        o.close()
    gg.close()


def test_pipeline_multistream(gpu):
    """streams are independent: a 3-stream lock-step batch equals three single-stream oracle encodes"""
    from gpu_enc import GpuEncoder
    w, h, S, nfr = 96, 80, 3, 3
    seqs = [synth_frames(w, h, nfr, seed=100 + s) for s in range(S)]
    gg = GpuEncoder(O.default_config(w, h, streams=S))
    ogs = [O.OracleEncoder(O.default_config(w, h)) for _ in range(S)]
    for i in range(nfr):
        st = 2 if i == 0 else 0
        g_mb, g_lv = gg.encode([seqs[s][i] for s in range(S)], st)
        for s in range(S):
            o_mb, o_lv = ogs[s].encode(seqs[s][i], st)
            compare(f"stream {s} frame {i}", (w + 15) // 16, g_mb[s], o_mb, g_lv[s], o_lv, gg.recon(s), ogs[s].recon())


@pytest.mark.parametrize("w,h,nfr,kw", [
    (1280, 720, 3, dict(dct8x8=1, partitions=7, refs=3)),        # BASELINE.json configs[1]: 720p, bit-exact
    (1280, 720, 4, dict(dct8x8=1, partitions=7, refs=3, chroma_me=1, mixed_refs=1)),      # ... with the bench's full medium toolset
    (1920, 1080, 2, dict(dct8x8=1, partitions=7, refs=3)),       # configs[2] geometry: 1088 coded rows, cropped output
    (1920, 1080, 3, dict(dct8x8=1, partitions=7, refs=3, chroma_me=1, mixed_refs=1)),     # the subme-5 toolset (bench.py --rd off)
    (1280, 720, 3, dict(dct8x8=1, partitions=7, refs=3, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63)),   # configs[1] with the headline toolset: RD + psy + trellis 1
    (1920, 1080, 3, dict(dct8x8=1, partitions=7, refs=3, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63)),  # configs[2]: the bench's headline toolset
    (3840, 2160, 2, dict(dct8x8=1, partitions=7, refs=2, chroma_me=1, mixed_refs=1, me_method=2, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=127, slices=33)),   # configs[3]: umh, --trellis 2 (slow's), slice threads
    (1920, 1080, 3, dict(dct8x8=1, partitions=7, refs=3, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63, slices=68, slices_plain=1)),  # configs[2], headline toolset, --slices 68 (one row each)
    (1920, 1080, 2, dict(dct8x8=1, partitions=7, refs=2, chroma_me=1, mixed_refs=1, aq_mode=1, qp_i=23, qp_p=26)),   # + variance AQ (CRF / ABR sessions)
    (3840, 2160, 2, dict(dct8x8=1, partitions=7, refs=2, chroma_me=1, mixed_refs=1, me_method=2, subme=9)),          # configs[3] geometry and toolset (slow: umh, subme 9)
])
def test_pipeline_bitexact_full_size(gpu, w, h, nfr, kw):
    """the headline geometries against the oracle (a few frames: the CPU oracle runs ~3 frames/s at 1080p)"""
    from gpu_enc import GpuEncoder
    frames = synth_frames(w, h, nfr, seed=w + h)
    cfg = O.default_config(w, h, **kw)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    mbw = (w + 15) // 16
    for i, f in enumerate(frames):
        st = 2 if i == 0 else 0
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"{w}x{h} frame {i}", mbw, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
    og.close(); gg.close()


def test_full_size_closed_loop_properties(gpu):
    """1080p, many streams in lock-step: size-independent properties instead of the (slow) oracle —
    every stream of a batch fed the same frames produces identical records/levels/recon (no cross-stream leakage),
    and the reconstruction stays close to the source (PSNR) for every frame of a short GOP."""
    from gpu_enc import GpuEncoder
    from synth import psnr
    w, h, S, nfr = 1920, 1080, 4, 4
    frames = synth_frames(w, h, nfr, seed=77)
    gg = GpuEncoder(O.default_config(w, h, streams=S, dct8x8=1, partitions=7, refs=3))
    for i, f in enumerate(frames):
        mb, lv = gg.encode([f] * S, 2 if i == 0 else 0)
        for s in range(1, S):
            assert np.array_equal(mb[s], mb[0]) and np.array_equal(lv[s], lv[0])
            assert np.array_equal(gg.recon(s), gg.recon(0))
        assert psnr(gg.recon(0)[:w * h], f[:w * h]) > 36.0
    gg.close()


def test_many_streams_long_gop_deterministic(gpu):
    """race / scheduling check for the wavefront kernels: 48 streams fed the SAME frames for a 24-frame GOP must stay
    identical to each other frame by frame (records, levels, reconstruction), and stream 0 must equal the oracle on the
    first frames.  Any ordering bug in the band / row hand-offs shows up as a stream that diverges."""
    from gpu_enc import GpuEncoder
    w, h, S, nfr = 352, 288, 48, 24
    frames = synth_frames(w, h, nfr, seed=4242)
    cfg = O.default_config(w, h, streams=S, dct8x8=1, partitions=7, refs=3)
    gg, og = GpuEncoder(cfg), O.OracleEncoder(O.default_config(w, h, dct8x8=1, partitions=7, refs=3))
    mbw = (w + 15) // 16
    for i, f in enumerate(frames):
        st = 2 if i == 0 else 0
        mb, lv = gg.encode([f] * S, st)
        same = [s for s in range(1, S) if not (np.array_equal(mb[s], mb[0]) and np.array_equal(lv[s], lv[0]))]
        assert not same, f"frame {i}: streams {same[:8]} diverge from stream 0"
        r0 = gg.recon(0)
        for s in (1, S // 2, S - 1):
            assert np.array_equal(gg.recon(s), r0), f"frame {i}: reconstruction of stream {s} differs"
        if i < 6:
            o_mb, o_lv = og.encode(f, st)
            compare(f"frame {i}", mbw, mb[0], o_mb, lv[0], o_lv, r0, og.recon())
    gg.close(); og.close()


def test_non_idr_intra_picture_keeps_the_references(gpu):
    """X264GPU_SLICE_I_NONIDR: an intra picture that does not empty the DPB — the P picture after it may (and here does) predict
    from pictures before it"""
    from gpu_enc import GpuEncoder
    w, h = 176, 144
    a, b = synth_frames(w, h, 4, seed=5), synth_frames(w, h, 1, seed=99)
    seq = [(a[0], 2), (a[1], 0), (a[2], 0), (b[0], 3), (a[3], 0), (a[2], 0)]            # the cut picture is followed by the old scene again
    cfg = O.default_config(w, h, refs=3, partitions=3)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    older = 0
    for i, (f, st) in enumerate(seq):
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"non-IDR I sequence frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
        if i == 4:
            older = int((o_mb["ref"][:, 0] >= 1).sum())
    assert older > 20, "the picture after the non-IDR I picture should reach behind it"


def test_external_mb_qp_offsets(gpu):
    """x264gpu_encoder_set_mb_qp_offsets: the lookahead's per-macroblock quantiser offsets (AQ - macroblock-tree, single floats) instead of
    the encoder's own AQ; NULL returns to the configured mode"""
    import torch
    from gpu_enc import GpuEncoder
    from x264vfw_amd import lib
    w, h = 208, 120
    frames = synth_frames(w, h, 4, seed=12)
    cfg = O.default_config(w, h, refs=2, partitions=7, dct8x8=1, chroma_me=1, qp_i=25, qp_p=28)
    og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
    rng = np.random.default_rng(5)
    n = og.n
    for i, f in enumerate(frames):
        off = None if i == 3 else (rng.integers(-1800, 1500, n) / np.float32(256) + rng.random(n, np.float32) / 64).astype(np.float32)
        og.set_mb_qp_offsets(off)
        d_off = None if off is None else torch.from_numpy(off[None].copy()).cuda()
        lib.check(lib.x264gpu_encoder_set_mb_qp_offsets(gg.h, None if d_off is None else d_off.data_ptr()), "set_mb_qp_offsets")
        st = 2 if i == 0 else 0
        o_mb, o_lv = og.encode(f, st)
        g_mb, g_lv = gg.encode([f], st)
        compare(f"external offsets frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
        if off is not None:
            assert len(np.unique(o_mb["qp"])) > 4


@pytest.mark.parametrize("aq", [0, 1])
def test_per_stream_quantisers(gpu, aq):
    """x264gpu_encoder_set_stream_qps: the streams of one call carry their own slice quantisers (GOP-parallel CRF); each stream must
    equal an oracle encoder driven with that quantiser alone, with and without AQ on top; NULL returns to the shared quantiser"""
    from gpu_enc import GpuEncoder
    from x264vfw_amd import lib
    w, h, S = 176, 144, 3
    seqs = [synth_frames(w, h, 4, seed=20 + s) for s in range(S)]
    cfg = O.default_config(w, h, refs=2, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=24, qp_p=27, aq_mode=aq)
    gcfg = O.default_config(w, h, streams=S, refs=2, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, qp_i=24, qp_p=27, aq_mode=aq)
    ogs, gg = [O.OracleEncoder(cfg) for _ in range(S)], GpuEncoder(gcfg)
    plan = [[20, 33, 27], [22, 38, 25], [30, 18, 44], None]                # per picture: a quantiser per stream, or back to the shared one
    for i in range(4):
        st = 2 if i == 0 else 0
        qps = plan[i]
        arr = None if qps is None else np.array(qps, np.int8)
        lib.check(lib.x264gpu_encoder_set_stream_qps(gg.h, None if arr is None else arr.ctypes.data), "set_stream_qps")
        g_mb, g_lv = gg.encode([seqs[s][i] for s in range(S)], st)
        for s in range(S):
            q = qps[s] if qps else (24 if st == 2 else 27)
            ogs[s].set_qp(q, q)
            o_mb, o_lv = ogs[s].encode(seqs[s][i], st)
            compare(f"stream qp aq={aq} frame {i} stream {s}", (w + 15) // 16, g_mb[s], o_mb, g_lv[s], o_lv, gg.recon(s), ogs[s].recon())
            if not aq:
                assert set(np.unique(o_mb["qp"])) == {q}


def test_lookahead_vectors_enter_the_16x16_candidates(gpu):
    """x264 hands the lookahead's vector of a macroblock (lowres_mvs[0][0], doubled) to the 16x16 search of reference 0 as its first
    candidate: x264gpu_encoder_set_lowres_mvs.  Useful vectors (the true global motion), useless ones and the "absent" marker, against
    the oracle fed the same arrays; the useful ones must actually change decisions somewhere"""
    import torch
    from gpu_enc import GpuEncoder
    w, h, nfr = 176, 144, 4
    base = synth_frames(w, h, 1, seed=31)[0]
    Y = base[:w * h].reshape(h, w)
    frames = []
    for i in range(nfr):                                   # a pan of 6 px per picture: the neighbours' vectors lag behind at the left edge
        yy = np.roll(Y, 6 * i, axis=1)
        frames.append(np.concatenate([yy.reshape(-1), base[w * h:]]))
    cfg = O.default_config(w, h, partitions=3, refs=2, subme=5, me_range=4)
    nmb = ((w + 15) // 16) * ((h + 15) // 16)
    rng = np.random.default_rng(5)
    recs = {}
    for name in ("none", "pan", "noise", "absent"):
        og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
        mv = None
        if name == "pan": mv = np.tile(np.array([-12, 0], np.int16), (nmb, 1))          # x 2 = quarter-pel: the content moved 6 px to the right
        elif name == "noise": mv = rng.integers(-20, 21, (nmb, 2)).astype(np.int16)
        elif name == "absent": mv = np.full((nmb, 2), 0x7fff, np.int16)
        d_mv = torch.from_numpy(mv).cuda() if mv is not None else None
        out = []
        for i, f in enumerate(frames):
            if mv is not None:
                O.L.x264o_encoder_set_lowres_mvs(og.h, O.ptr(mv))
                gpu.check(gpu.x264gpu_encoder_set_lowres_mvs(gg.h, d_mv.data_ptr()))
            o_mb, o_lv = og.encode(f, 2 if i == 0 else 0)
            g_mb, g_lv = gg.encode([f], 2 if i == 0 else 0)
            compare(f"lowres mvs {name} frame {i}", (w + 15) // 16, g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon())
            out.append(o_mb.copy())
        recs[name] = out
        og.close(); gg.close()
    same = lambda a, b: all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b))
    assert same(recs["none"], recs["absent"])
    assert not same(recs["none"], recs["pan"])

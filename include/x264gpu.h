/* include/x264gpu.h — boundary B3: thin C ABI between the host encoder (behind x264_encoder_encode,
 * reference call site codec.c:1693) and the hand-written HIP kernels for gfx950.
 *
 * Plain C: pointers + sizes only, no C++/torch types.  All `d_` pointers are device (HBM) pointers,
 * `stream` is a hipStream_t passed as void* (NULL = default stream).  Every function returns 0 on
 * success or a negative X264GPU_E* code; there is NO CPU fallback — without a GPU the calls fail.
 *
 * Each entry point names the libx264 internal it replaces — libx264 itself is not
 * in the reference tree, so the citation is the reference call site plus the upstream location.
 */
#ifndef X264GPU_H
#define X264GPU_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X264GPU_ABI_VERSION 2
enum { X264GPU_OK = 0, X264GPU_EINVAL = -1, X264GPU_EHIP = -2, X264GPU_ENOMEM = -3, X264GPU_ENODEV = -4 };

/* ------------------------------------------------------------------------------------------------
 * runtime / memory (so C callers and ctypes tests need nothing but this library)
 * ---------------------------------------------------------------------------------------------- */
int  x264gpu_abi_version(void);
int  x264gpu_device_count(void);               /* 0 when no GPU is visible */
int  x264gpu_set_device(int dev);
int  x264gpu_get_device(int *dev);        /* the calling thread's device: a helper thread that issues work for an encoder selects it first */
const char *x264gpu_last_error(void);
int  x264gpu_malloc(void **d_ptr, size_t bytes);
int  x264gpu_free(void *d_ptr);
int  x264gpu_memcpy_h2d(void *d_dst, const void *h_src, size_t bytes, void *stream);
int  x264gpu_memcpy_d2h(void *h_dst, const void *d_src, size_t bytes, void *stream);
int  x264gpu_memcpy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);   /* asynchronous on `stream` */
int  x264gpu_memset(void *d_dst, int value, size_t bytes, void *stream);
int  x264gpu_stream_sync(void *stream);
/* A stream of the caller's own (it does not wait for the default stream, nor the default stream for it): what the `stream` arguments of this header take besides
 * NULL.  The host encoder's batcher downloads one round's records on one while the next round runs on the default stream. */
int  x264gpu_stream_create(void **stream);
int  x264gpu_stream_destroy(void *stream);
/* Events: "everything issued on `stream` before the record is done" as something a host thread can wait for WITHOUT waiting for what was issued after it
 * (x264gpu_stream_sync waits for the whole stream).  The batcher records one behind every round: the round's downloads wait for theirs while the next round
 * is already queued on the same stream. */
int  x264gpu_event_create(void **event);
int  x264gpu_event_destroy(void *event);
int  x264gpu_event_record(void *event, void *stream);
int  x264gpu_event_sync(void *event);
int  x264gpu_stream_wait_event(void *stream, void *event);   /* what is issued on `stream` after this call runs after the event (device side: the host does not wait) */

/* ------------------------------------------------------------------------------------------------
 * Tier 1 — DSP primitives in batch form (the "checkasm" surface: same device code the frame
 * pipeline uses, exposed so every primitive is parity-tested against oracle/ in isolation).
 * Block arrays are tightly packed: block i of a WxH batch starts at base + i*W*H, row stride W.
 * ---------------------------------------------------------------------------------------------- */
/* A2/A3: pixel_sad_WxH, pixel_satd_WxH, pixel_sa8d_{8x8,16x16}, pixel_ssd, pixel_var
 * ([x264-upstream] common/pixel.c).  metric: 0 SAD, 1 SATD, 2 SA8D, 3 SSD.  W,H in {4,8,16}. */
int x264gpu_pixel_metric(int metric, const uint8_t *d_a, const uint8_t *d_b, int n, int w, int h,
                         int32_t *d_out, void *stream);
/* pixel_var_{16x16,8x8}: out[i] = sum | (sqr << 32) */
int x264gpu_pixel_var(const uint8_t *d_a, int n, int w, int h, uint64_t *d_out, void *stream);
/* pixel_hadamard_ac_{16x16,16x8,8x16,8x8} (psy-RD energy, A3 / A11): out[i] = ((sum8 >> 2) << 32) | (sum4 >> 1), DC terms removed */
int x264gpu_pixel_hadamard_ac(const uint8_t *d_a, int n, int w, int h, uint64_t *d_out, void *stream);

/* A6/A7/A8: sub4x4_dct -> quant_4x4 -> (levels out) -> dequant_4x4 -> add4x4_idct, per 4x4 block
 * ([x264-upstream] common/dct.c, common/quant.c).  enc/pred: n blocks of 4x4 u8.  Outputs (any may be
 * NULL): coef = forward transform, levels = quantised (raster order), recon = reconstructed pixels.
 * list: 0 intra-luma 1 inter-luma 2 intra-chroma 3 inter-chroma (deadzone lists). */
int x264gpu_dctq4x4(const uint8_t *d_enc, const uint8_t *d_pred, int n, int qp, int list,
                    int16_t *d_coef, int16_t *d_levels, uint8_t *d_recon, void *stream);
/* 8x8 transform path: sub8x8_dct8 -> quant_8x8 -> dequant_8x8 -> add8x8_idct8.  list: 0 intra 1 inter */
int x264gpu_dctq8x8(const uint8_t *d_enc, const uint8_t *d_pred, int n, int qp, int list,
                    int16_t *d_coef, int16_t *d_levels, uint8_t *d_recon, void *stream);

/* A5: intra predictors ([x264-upstream] common/predict.c; H.264 8.3).  d_plane is a reconstructed
 * plane (stride bytes); block i sits at pixel (xy[2i], xy[2i+1]); out = n tightly packed predictions.
 * kind: 0 = 16x16 luma, 1 = 8x8 chroma, 2 = 4x4 luma.  avail: X264O_AVAIL_* style bits (4x4 only). */
int x264gpu_intra_predict(int kind, const uint8_t *d_plane, int stride, const int32_t *d_xy,
                          const int32_t *d_mode, const int32_t *d_avail, int n, uint8_t *d_out, void *stream);

/* A4: hpel_filter + border expansion ([x264-upstream] common/mc.c hpel_filter, common/frame.c
 * x264_frame_filter / frame_expand_border).  d_planes = 4 contiguous padded planes (full,H,V,HV), each
 * plane_bytes apart; pointers address the padded origin; sample (0,0) is at pad*stride+pad.
 * Reads plane 0 interior (w x h), writes all of planes 1..3 and the border of plane 0. */
int x264gpu_hpel_filter(uint8_t *d_planes, size_t plane_bytes, int stride, int w, int h, int pad, void *stream);
/* A1: frame_init_lowres_core ([x264-upstream] common/mc.c): 4 half-res planes, dst stride ds */
int x264gpu_lowres(const uint8_t *d_src, int ss, int w, int h, uint8_t *d_dst, size_t plane_bytes, int ds,
                   void *stream);
/* A10: mc_luma / get_ref and mc_chroma ([x264-upstream] common/mc.c).  Block i of size w x h at
 * (xy[2i],xy[2i+1]) with quarter-pel mv (mv[2i],mv[2i+1]); luma reads the 4 padded planes, chroma
 * reads a padded NV12 plane (d_ref addresses sample (0,0)) and writes U then V blocks. */
int x264gpu_mc_luma(const uint8_t *d_planes00, size_t plane_bytes, int stride, const int32_t *d_xy,
                    const int32_t *d_mv, int n, int w, int h, uint8_t *d_out, void *stream);
int x264gpu_mc_chroma(const uint8_t *d_nv12_00, int stride, const int32_t *d_xy, const int32_t *d_mv,
                      int n, int w, int h, uint8_t *d_out, void *stream);
/* A10: the two sample combiners behind bi-prediction and --weightp ([x264-upstream] common/mc.c pixel_avg_wxh / pixel_avg_weight_wxh and
 * mc_weight), on `bytes` already motion-compensated samples (a multiple of 4; any block shape, blocks contiguous):
 * avg: weight1 == 32 -> (a + b + 1) >> 1, else clip((a * weight1 + b * (64 - weight1) + 32) >> 6);
 * weight: clip(((src * scale + (1 << (denom - 1))) >> denom) + offset), denom 0: clip(src * scale + offset). */
int x264gpu_mc_avg(const uint8_t *d_a, const uint8_t *d_b, size_t bytes, int weight1, uint8_t *d_out, void *stream);
/* x264's CABAC trellis quantiser (--trellis 1; [x264-upstream] encoder/rdo.c quant_trellis_cabac) as a primitive: nblk blocks of block category
 * cat (0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8), coefficients in scan order (16 per block; 64 for cat 5, 4 for
 * cat 3; AC blocks: entry 0 unused = 0), quantiser qp, inter / intra lambda, against the 460 context variables ((pStateIdx << 1) | valMPS) of a
 * slice.  d_levels receives the levels in the same layout, d_nz one byte per block.  The macroblock loop does not use it yet (cfg.trellis). */
int x264gpu_trellis_blocks(const int16_t *d_coefs, int nblk, int cat, int qp, int intra, const uint8_t *d_states460, int16_t *d_levels, uint8_t *d_nz, void *stream);
int x264gpu_mc_weight(const uint8_t *d_src, size_t bytes, int scale, int denom, int offset, uint8_t *d_out, void *stream);
/* The level walk of the CABAC size pricing as a primitive ([x264-upstream] encoder/cabac.c coeff_abs_level_minus1 of residual_block_cabac, every block of a
 * macroblock at once — csrc/cabac_rd.hip.h cab_levels_all): n cases; d_levels the macroblocks' levels in x264gpu_mb layout (X264GPU_MB_LEVELS each);
 * d_what five ints a case: luma category (2 = 4x4, 5 = 8x8, 1 = Intra_16x16 AC, -1 none), mask of luma blocks, of chroma AC blocks (plane * 4 + block), of
 * chroma DC planes, luma DC flag; d_r / d_r8 the role-indexed context registers (64 lanes a case, csrc/cabac_layout.hip.h) in, d_r_out / d_r8_out out;
 * d_bits the bits in 1/256 (sign and escape bypass bins included).  tests/test_gpu_prims.py checks it against a serial restatement. */
int x264gpu_cabac_level_walk(const int16_t *d_levels, const int32_t *d_what, int n, const uint32_t *d_r, const uint32_t *d_r8, uint32_t *d_r_out, uint32_t *d_r8_out, int32_t *d_bits, void *stream);

/* ---- input colourspace conversion to I420 -----------------------------------------
 * Replaces the x264vfw_csp_function_t table the driver installs for an I420 encoder (/root/reference/csp.c:436-487,
 * called at codec.c:1799): plane copies (I420, YV12 with the U/V swap), YV16 / YV24 subsampling (csp.c:39-73),
 * YUY2 / UYVY de-interleave (csp.c:155-205) and BGR / BGRA in 20-bit fixed point for BT.601/709 x TV/PC range
 * (csp.c:252-388), each with the optional vertical flip.  csp ids are the driver's (csp.h:30-44). */
enum { X264GPU_CSP_MASK = 0xff, X264GPU_CSP_I420 = 1, X264GPU_CSP_YV12 = 2, X264GPU_CSP_YV16 = 3, X264GPU_CSP_YV24 = 4,
       X264GPU_CSP_NV12 = 5, X264GPU_CSP_YUYV = 6, X264GPU_CSP_UYVY = 7, X264GPU_CSP_BGR = 8, X264GPU_CSP_BGRA = 9,
       X264GPU_CSP_VFLIP = 0x1000 };
/* plane offsets / strides of a frame of `csp` held in one contiguous buffer (x264vfw_img_fill, codec.c:304-379);
 * returns the buffer size in bytes, or -1 for an unknown csp */
long x264gpu_csp_img_fill(int csp, int width, int height, long off[3], int stride[3]);
/* device -> device conversion; width/height even; colmatrix709 / fullrange select the RGB matrix exactly as
 * x264vfw_csp_init(i_colmatrix == 1, b_fullrange) does.  Packed formats use d_src[0] only. */
int x264gpu_csp_to_i420(const uint8_t *const d_src[3], const int src_stride[3], int csp, int width, int height,
                        int colmatrix709, int fullrange, uint8_t *const d_dst[3], const int dst_stride[3], void *stream);
/* the same for `frames` pictures of identical geometry laid out src_frame_bytes / dst_frame_bytes apart: one launch per
 * plane group instead of one per picture (a 1080p picture is ~3 us of HBM time, launch overhead would dominate) */
int x264gpu_csp_to_i420_batch(const uint8_t *const d_src[3], const int src_stride[3], size_t src_frame_bytes, int csp, int width,
                              int height, int colmatrix709, int fullrange, uint8_t *const d_dst[3], const int dst_stride[3],
                              size_t dst_frame_bytes, int frames, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Tier 2 — frame pipeline: the hot path of x264_encoder_encode for a batch of independent
 * closed-GOP streams (one launch covers `streams` frames of identical geometry).
 * ---------------------------------------------------------------------------------------------- */
enum { X264GPU_MB_I4x4 = 0, X264GPU_MB_I8x8 = 1, X264GPU_MB_I16x16 = 2, X264GPU_MB_P_L0 = 4, X264GPU_MB_P_8x8 = 5,
       X264GPU_MB_P_SKIP = 6,
       /* B slices.  B_DIRECT / B_SKIP: spatial direct prediction of the whole macroblock (with / without residual), the derived references
        * and vectors of both lists are in the record.  B_INTER: partition 16x16 / 16x8 / 8x16, every partition predicted from list 0
        * (ref[k] >= 0, ref1[k] < 0), list 1 (ref[k] < 0, ref1[k] >= 0) or both — x264's B_L0_L0 .. B_BI_BI.  B_8x8: partition 8x8, each
        * 8x8 block L0 / L1 / BI by the same rule or direct (bit k of direct8) */
       X264GPU_MB_B_DIRECT = 7, X264GPU_MB_B_SKIP = 8, X264GPU_MB_B_INTER = 9, X264GPU_MB_B_8x8 = 10 };
enum { X264GPU_SLICE_P = 0, X264GPU_SLICE_B = 1, X264GPU_SLICE_I = 2 /* IDR picture: empties the DPB */,
       X264GPU_SLICE_I_NONIDR = 3 /* intra picture that keeps the DPB (x264's X264_TYPE_I, e.g. a scenecut inside min-keyint) */ };

/* per-macroblock decision record written by the GPU, consumed by the host entropy coder (64 B) */
typedef struct x264gpu_mb {
    uint8_t  type;          /* X264GPU_MB_* */
    union {
    uint8_t  i16_mode;      /* intra 16x16 prediction mode (I_PRED_16x16_*, real modes 0..3) */
    uint8_t  direct8;       /* B_8x8: bit k = 8x8 block k is a direct sub-macroblock (B_DIRECT / B_SKIP: 15) */
    };
    uint8_t  chroma_mode;   /* intra chroma prediction mode 0..3 */
    uint8_t  qp;            /* luma qp used for this macroblock */
    uint8_t  cbp_luma;      /* bit i = 8x8 block i has coded coefficients */
    uint8_t  cbp_chroma;    /* 0 none, 1 DC only, 2 DC+AC */
    uint8_t  partition;     /* 0 16x16, 1 16x8, 2 8x16, 3 8x8 (P only) */
    int8_t   ref[4];        /* reference index per 8x8 (list 0; -1 = the block does not use the list) */
    union {
    uint8_t  i4_mode[16];   /* intra 4x4 modes, x264 block order (zigzag-of-8x8); I8x8: mode of 8x8 block i in [4i..4i+3] */
    int8_t   ref1[4];       /* inter macroblocks of B slices: list-1 reference index per 8x8 (-1 = the block does not use the list) */
    };
    uint8_t  transform8x8;  /* transform_size_8x8_flag: luma residual uses the 8x8 transform (0 when cbp_luma == 0) */
    int16_t  mv[4][2];      /* quarter-pel motion vector per 8x8 (x,y), list 0 */
    uint32_t nnz;           /* bit b (0..15 luma blk order [transform8x8: of the interleaved 4x4s], 16..19 U, 20..23 V, 24 lumaDC, 25 U DC, 26 V DC) */
    union {
    struct {
    int32_t  cost;          /* analysis cost of the chosen mode (diagnostic) */
    int32_t  aux[3];        /* diagnostics: [0] best inter cost, [1] intra-16x16 source estimate (P slices) */
    };
    int16_t  mv1[4][2];     /* inter macroblocks of B slices: list-1 vector per 8x8, in place of the diagnostics */
    };
} x264gpu_mb;

/* quantised levels per macroblock, scan (zigzag) order, int16:
 *   [0..255]   16 luma 4x4 blocks x 16 (x264 block order; for I16x16 index 0 of each block is unused/0).
 *              transform8x8: 8x8 block i occupies blocks 4i..4i+3 in CAVLC-interleaved form, i.e. level z of
 *              the 8x8 zigzag sits at block 4i + (z & 3), index z >> 2 (x264 zigzag_interleave_8x8_cavlc)
 *   [256..271] luma DC (I16x16 only)
 *   [272..279] chroma DC: U[4], V[4]
 *   [280..407] chroma AC: 8 blocks x 16 (index 0 of each unused/0)
 *   [408..415] padding   -> 416 int16 = 832 bytes */
#define X264GPU_MB_LEVELS 416
#define X264GPU_LV_LUMA 0
#define X264GPU_LV_LUMA_DC 256
#define X264GPU_LV_CHROMA_DC 272
#define X264GPU_LV_CHROMA_AC 280

typedef struct x264gpu_encoder x264gpu_encoder;  /* opaque: owns the device-resident DPB + work buffers */

typedef struct x264gpu_config {
    int width, height;        /* luma picture size (even) */
    int streams;              /* independent streams/closed GOPs encoded in lock-step */
    int refs;                 /* reference frames for P slices, 1..5 */
    int qp_i, qp_p;           /* constant QPs (X264_RC_CQP path, codec.c:1498-1502) */
    int me_range;             /* --merange (16) */
    int subme;                /* --subme level 0..9: the sub-pel iteration table, SATD from 2 up, chroma-ME from 5 (B slices: 9), and with `rd` the RD levels
                               * (6 / 7 mode decision, 8 / 9 refinement; B slices analyse one level down: without RD below 7).  With rd != 0, subme >= 8 needs
                               * cabac and --me hex / umh (the 4 + 10 sub-pel iterations live in the refinement instantiations); with B pictures so does 9 */
    int deblock;              /* 1 = in-loop filter on */
    int deblock_alpha, deblock_beta; /* --deblock a:b offsets */
    int chroma_qp_offset;
    int deadzone_inter, deadzone_intra;
    int dct_decimate;
    int partitions;           /* P slices: bit0 p8x8 (16x8/8x16/8x8), bit1 i4x4, bit2 i8x8 (needs dct8x8); with bit8 set B slices take their 8x8 / 16x8 / 8x16
                               * analysis from bit11 (--partitions b8x8) instead of bit0.  I slices use the
                               * same bits 1-2 unless bit8 is set, then bit9 = i4x4 and bit10 = i8x8 (x264 keeps separate
                               * analyse.intra / analyse.inter masks) */
    int dct8x8;               /* --8x8dct: adaptive 8x8 luma transform (High profile) */
    int me_method;            /* --me: 0 dia (radius-1 diamond), 1 hex (hexagon + square refine), 2 umh (uneven multi-hexagon), 3 esa (exhaustive); X264_ME_DIA / _HEX / _UMH / _ESA */
    int chroma_me;            /* --chroma-me (x264 default on): sub-pel SATD costs of P macroblocks include the chroma planes; acts when subme >= 5,
                               * as x264's h->mb.b_chroma_me ([x264-upstream] encoder/encoder.c, me.c COST_MV_SATD) */
    int mixed_refs;           /* --mixed-refs (x264 default on): 8x8 blocks, and the 16x8 / 8x16 halves built on them, choose their reference
                               * on their own ([x264-upstream] analyse.c x264_mb_analyse_inter_p8x8_mixed_ref); needs partitions bit0 and refs > 1 */
    int aq_mode;              /* --aq-mode: 0 off, 1 variance AQ (x264_adaptive_quant_frame, [x264-upstream] encoder/ratecontrol.c): every macroblock's
                               * quantiser = clip3((int)(qpm + strength * (x264_log2(max(energy, 1)) - 14.427f) + 0.5f)), energy = AC energy of its luma
                               * and chroma source samples, qpm the picture's float quantiser (x264gpu_pic.qpm) — x264's own single-float expressions
                               * (x264_log2's table: include/x264gpu_log2f_lut.inc), evaluated without contraction on host checker and device alike.
                               * x264 switches AQ off under constant QP; so does the host encoder. */
    float aq_strength;        /* --aq-strength * 1.0397f (x264_adaptive_quant_frame's strength of mode 1; x264 default 1.0 -> 1.0397f) */
    int fast_pskip;           /* --no-fast-pskip clears it (x264 default on): P_Skip probed inside the analysis (x264_macroblock_probe_pskip) */
    int mv_range;             /* --mvrange in luma samples, both directions; 0 = 512.  x264 takes it from the level (x264_levels[].mv_range) */
    int cabac;                /* the session's entropy coder is CABAC (x264 b_cabac).  Entropy coding stays on the host, but x264's analysis knows the coder:
                               * under CABAC the P8x8 sub-macroblock type costs nothing (no sub-8x8 analysis here) and reference 0 of a P8x8 macroblock is
                               * costed like any other; under CAVLC P_8x8ref0 makes it free ([x264-upstream] analyse.c x264_mb_analyse_inter_p8x8*) */
    int rd;                   /* 1: x264's RD mode decision of subme 6 / 7 (i_mbrd 1: x264_rd_cost_mb = SSD + psy + lambda2 x bits over the candidate
                               * macroblock types, transform-size RD, P_SKIP by RD; no final quarter-pel refinement).  Bits: exact CAVLC counts
                               * (cabac == 0), or x264's size-only CABAC on the slice's context variables, which the device carries (cabac == 1).
                               * 1 | sites << 1 with subme 8 (CABAC sessions, --me hex / umh): RD refinement of the chosen type as well (x264 i_mbrd 2) —
                               * sites: 1 the vectors of the P partitions (x264_me_refine_qpel_rd), 2 the Intra_16x16 mode, 4 the chroma mode, 8 the
                               * Intra_4x4 modes, 16 the Intra_8x8 modes (intra_rd_refine); x264's --subme 8 is all of them: rd = 63.  At subme 9 site 1 also
                               * covers B slices (x264_me_refine_qpel_rd per list, x264_me_refine_bidir_rd) and the intra sites their intra macroblocks.
                               * + 64: deblock-aware RD (x264 h->mb.b_deblock_rdo, --subme 9 with the loop filter on): whole-macroblock RD candidates are
                               * measured after x264_macroblock_deblock; x264's --subme 9 = 63 | 64 (cabac, --me hex / umh) */
    int psy;                  /* x264 analyse.b_psy (default on): chroma lambda offset of the RD costs */
    int psy_rd_q8;            /* x264 FIX8(--psy-rd strength) (medium: 256); enters the RD costs (subme >= 6) */
    int slices;               /* x264 --sliced-threads with --threads N: N slices per picture (0 / 1 = one), slice i = macroblock rows
                               * [(mbh * i + N/2) / N, (mbh * (i+1) + N/2) / N); every slice is analysed on its own (no prediction across a slice
                               * boundary, its own fast-intra statistics and quantiser chain) and the loop filter leaves slice boundaries alone
                               * (disable_deblocking_filter_idc 2), as x264's slice threads do.  At most mbh / 4 ([x264-upstream] validate_parameters);
                               * see slices_plain for --slices N */
    int trellis;              /* x264's trellis quantiser (quant_trellis_cabac: a search over the levels of a block on the slice's live CABAC state); needs cabac and
                               * rd.  Bits 0..5 = the quantiser calls that use it (1 inter luma 4x4, 2 inter luma 8x8, 4 chroma, 8 Intra_16x16, 16 Intra_4x4,
                               * 32 Intra_8x8): --trellis 1 = 63, in the final encode of every macroblock.  Bit 6 (64) = --trellis 2: also in the block encodes of
                               * the intra analysis and in every RD candidate */
    int slices_plain;         /* 1: `slices` are x264's --slices N (i_slice_count) instead of its slice threads: same row split, no prediction across a
                               * boundary, but (a) the loop filter runs ACROSS slice boundaries (disable_deblocking_filter_idc 0, [x264-upstream]
                               * slice_header_init: 2 only under b_sliced_threads), (b) a slice may be a single macroblock row (at most mbh slices) and
                               * (c) x264 codes these slices one after the other in one thread, so the frame statistics its fast-intra decision reads
                               * (intra macroblocks so far) run on through the picture.  The slices still run side by side on the device: each on an
                               * assumed count of the slices before it, and those whose decisions hang on a wrong assumption run again */
    int dpb;                  /* reference pictures the DPB holds (x264 sps->i_num_ref_frames: max(refs, 4 under --b-pyramid, ...)); 0 = refs.  The encoder
                               * owns dpb + 1 picture slots: x264gpu_encode_pictures names the slot every picture is reconstructed into */
    int weightb;              /* x264 --weightb (default on): implicit weighted bi-prediction, weights from the POC distances (weighted_bipred_idc 2) */
} x264gpu_config;

/* One picture of one stream for x264gpu_encode_pictures: slice type, quantiser and the reference lists as DPB slots — what x264's
 * x264_reference_build_list hands the slice ([x264-upstream] encoder/encoder.c): list 0 = pictures before it in display order, nearest
 * first; list 1 (B) = pictures after it; the host keeps the DPB (sliding window, --b-pyramid, duplicates of --weightp) and names slots. */
#define X264GPU_MAX_LIST 8
typedef struct x264gpu_pic {
    int slice_type;           /* X264GPU_SLICE_I (IDR) / _I_NONIDR / _P / _B — the same for every stream of a call */
    int qp;                   /* slice quantiser */
    int poc;                  /* picture order count: 2 x (display index since the IDR) */
    int dst;                  /* DPB slot that receives the reconstruction (0 .. dpb) */
    int keep;                 /* 1: the picture will be referenced (half-pel planes and borders are built); 0: a non-reference b */
    int nref[2];              /* active references of list 0 / list 1 */
    int8_t slot[2][X264GPU_MAX_LIST];      /* DPB slot of reference index i of list l */
    /* P pictures, x264 --weightp: explicit luma weight of list-0 index i (8.4.2.3.2): clip(((p * scale + (1 << (denom - 1))) >> denom) + offset),
     * denom 0: clip(p * scale + offset); on == 0: the index is not weighted.  x264's --weightp 2 puts a DUPLICATE of reference 0 at index 1 with
     * weight {scale 1, denom 0, offset -1} on every P picture that has at least two references (blind_dupe = that index, else -1): the host names
     * the same slot twice in slot[0][], the analysis refines the duplicate from reference 0's vector instead of searching it */
    struct { int8_t on, denom; int16_t scale, offset; } wl0[X264GPU_MAX_LIST];
    int blind_dupe;
    float qpm;                /* rate-controlled sessions: the picture's FLOAT quantiser (x264 rc->qpm; qp is its rounding): a macroblock's quantiser under AQ /
                               * macroblock-tree is clip3((int)(qpm + its offset + 0.5f)) as x264_ratecontrol_mb_qp has it.  0 = (float)qp */
    /* ... and the explicit chroma weights of list-0 index i (x264_weights_analyse weights the chroma planes of a fade once luma got a weight):
     * plane 0 = Cb, 1 = Cr, one denominator for both (chroma_log2_weight_denom); same formula as luma */
    struct { int8_t on[2], denom, pad; int16_t scale[2], offset[2]; } wc0[X264GPU_MAX_LIST];
    /* B pictures, x264 --direct: direct_temporal = 1: temporal direct prediction (8.4.1.2.3; direct_spatial_mv_pred_flag 0), 0: spatial.  This one
     * field may differ between the streams of a call (each stream's --direct auto state decides it).  direct_auto = 1 (--direct auto, every
     * stream alike): each macroblock also predicts with the OTHER mode and counts which of the two would let x264_macroblock_probe_bskip skip it
     * (h->stat.frame.i_direct_score[]; x264gpu_encoder_direct_scores) — the host feeds the counts into the next B picture's choice */
    int direct_temporal, direct_auto;
} x264gpu_pic;

/* --direct auto: h_scores[streams][2] = how many macroblocks of the last B picture coded with direct_auto the skip probe passed under temporal [0] /
 * spatial [1] direct prediction (x264 h->stat.frame.i_direct_score) */
int  x264gpu_encoder_direct_scores(x264gpu_encoder *enc, int *h_scores);
int  x264gpu_encoder_create(x264gpu_encoder **enc, const x264gpu_config *cfg);
/* A second LAUNCH CONTEXT over `parent`'s DPB (x264gpu_encode_pictures sessions): the picture slots and their side data are the parent's, the per-launch scratch
 * the view's own.  Pictures of one session that share only finished references — the b pictures of a mini-GOP, the next P picture, the B reference between two
 * finished P pictures: x264's frame threads code them side by side too — can be issued through different contexts on different streams and run together; the
 * CALLER orders them: a picture is issued after the events of the pictures it references (x264gpu_stream_wait_event) and into a slot no picture in flight reads
 * or writes.  The results are what the same calls one after the other through `parent` give.  Destroy views before the parent. */
int  x264gpu_encoder_create_view(x264gpu_encoder **enc, x264gpu_encoder *parent);
void x264gpu_encoder_destroy(x264gpu_encoder *enc);
/* sizes of the per-frame outputs for ONE stream */
int  x264gpu_encoder_mb_count(const x264gpu_encoder *enc);
/* Encode one frame per stream.  d_i420: `streams` tightly packed I420 pictures (W*H*3/2 each) already
 * resident in HBM.  slice_type: X264GPU_SLICE_I (IDR, resets the DPB), X264GPU_SLICE_I_NONIDR or X264GPU_SLICE_P.
 * Outputs stay on the device: d_mb [streams][mb_count] records, d_levels [streams][mb_count][416].
 * Replaces (for the slice types supported) x264_frame_copy_picture, x264_macroblock_analyse,
 * x264_macroblock_encode, x264_frame_deblock_row, x264_frame_filter of [x264-upstream]. */
int  x264gpu_encode_frames(x264gpu_encoder *enc, const uint8_t *d_i420, int slice_type,
                           x264gpu_mb *d_mb, int16_t *d_levels, void *stream);
/* The same with explicit picture control, one entry of `pics` (host array) per stream: B pictures, b-pyramid, per-stream quantisers.
 * The streams of a call run in LOCK-STEP: slice type, POC, destination slot, keep flag, both reference lists, explicit weights and the
 * blind duplicate must be the same in every entry (EINVAL otherwise); qp and qpm may differ per stream.  B pictures need a CABAC
 * session with RD (cfg.cabac, cfg.rd).  x264gpu_encode_frames is this call with the sliding-window DPB of an I / P stream. */
int  x264gpu_encode_pictures(x264gpu_encoder *enc, const uint8_t *d_i420, const x264gpu_pic *pics,
                             x264gpu_mb *d_mb, int16_t *d_levels, void *stream);
/* The levels of a call, packed IN PLACE for the trip to the host: of every macroblock's X264GPU_MB_LEVELS levels only the groups of 16 that hold a non-zero one stay,
 * one behind the other from the start of the stream's array (a P picture at medium keeps ~10 % of the bytes, a B picture ~3 %, an I picture ~25 %).
 * d_index[streams][mb_count] tells where: `at` = the macroblock's first kept group, counted in groups from the start of the stream's array; bit g of `groups` = levels
 * [16 g, 16 g + 16) are kept (the kept groups follow each other in g order).  The bytes of stream s that matter are its first (at + popcount(groups)) x 32 of the
 * last macroblock.  The host's slice writers take the pair (host/slice.cpp LevelSource).  x264_macroblock_write_* of [x264-upstream] reads h->dct the same way:
 * what was not coded is not looked at. */
typedef struct x264gpu_level_index { uint32_t at, groups; } x264gpu_level_index;
int  x264gpu_pack_levels(int16_t *d_levels, int streams, int mb_count, x264gpu_level_index *d_index, uint32_t *d_kept /* [streams]: groups kept per stream, or NULL */, void *stream);
/* A9 as a primitive ([x264-upstream] common/deblock.c x264_frame_deblock_row over a whole picture): the in-loop filter alone on
 * `streams` given pictures (d_i420: I420, width and height multiples of 16) with given macroblock records, through the kernel the
 * frame pipeline launches; alpha / beta / chroma-qp offsets from the encoder's configuration.  d_out: the filtered pictures. */
int x264gpu_encoder_deblock_pictures(x264gpu_encoder *e, const uint8_t *d_i420, const x264gpu_mb *d_mb, uint8_t *d_out, void *stream);
/* copy the reconstructed (deblocked) picture of `stream_idx` out as I420 (for parity tests / PSNR) */
int  x264gpu_encoder_get_recon(x264gpu_encoder *enc, int stream_idx, uint8_t *d_i420_out, void *stream);
/* the same for the picture in DPB slot `slot` (x264gpu_encode_pictures sessions) */
int  x264gpu_encoder_get_recon_slot(x264gpu_encoder *enc, int stream_idx, int slot, uint8_t *d_i420_out, void *stream);
/* Per-stage device timing for bench.py (HIP events on the caller's stream, no host sync in the timed
 * region): profile_begin arms up to max_calls encode_frames calls; profile_end synchronises the stream
 * and returns, per stage, the summed milliseconds and the number of launches that ran. */
int  x264gpu_encoder_stage_count(void);
const char *x264gpu_encoder_stage_name(int i);
int  x264gpu_encoder_profile_begin(x264gpu_encoder *enc, int max_calls);
/* diagnostics (NULL = off, the default): d_counters = [streams][16 waves][16] uint64 cycle counters written by the
 * wavefront kernels: {intra wait, work, MBs, total, deblock wait, work, MBs, total, 5 intra section sums, 3 spare} */
int  x264gpu_encoder_set_debug(x264gpu_encoder *enc, void *d_counters);
int  x264gpu_encoder_profile_end(x264gpu_encoder *enc, void *stream, double *ms_sum, int *launches);
/* Quantisers of the following x264gpu_encode_frames calls (every stream of the call shares them): what x264_ratecontrol_start
 * hands the slice ([x264-upstream] encoder/ratecontrol.c; CRF / scenecut sessions change it per picture). */
int  x264gpu_encoder_set_qp(x264gpu_encoder *enc, int qp_i, int qp_p);
/* Per-macroblock quantiser offsets (single floats, [streams][mb_count]) for the following pictures, as decided by the lookahead (x264: frame->f_qp_offset
 * = AQ - macroblock-tree, or f_qp_offset_aq; applied by x264_ratecontrol_mb_qp): quantiser = (int)(qpm + offset + 0.5f), clipped to 1..51.  The device array
 * must stay valid until the encode that uses it has been issued; NULL returns to the encoder's own aq_mode. */
int  x264gpu_encoder_set_mb_qp_offsets(x264gpu_encoder *enc, const float *d_offsets);
/* One slice quantiser per stream (host array of `streams` values, 0..51) for the following pictures, instead of the shared one of
 * x264gpu_encoder_set_qp: the streams of a call are then rate-controlled independently (GOP-parallel coding of one sequence under
 * CRF: every GOP slot carries the quantiser x264_ratecontrol_start would have given that picture).  AQ / caller offsets add to it per
 * macroblock.  NULL returns to the shared quantiser. */
int  x264gpu_encoder_set_stream_qps(x264gpu_encoder *enc, const int8_t *qps);
/* ... with every stream's FLOAT quantiser beside it (x264 rc->qpm; qps[s] is its rounding; qpms NULL or an entry 0: the integer one) */
int  x264gpu_encoder_set_stream_qpms(x264gpu_encoder *enc, const int8_t *qps, const float *qpms);
/* The float quantiser (x264 rc->qpm) of the following x264gpu_encode_frames pictures, every stream alike; the integer quantisers of x264gpu_encoder_set_qp
 * must be its rounding.  0 (the default) = the integer quantiser as a float.  x264gpu_encode_pictures carries it in x264gpu_pic.qpm instead. */
int  x264gpu_encoder_set_qpm(x264gpu_encoder *enc, float qpm);
/* Lookahead vectors of the following pictures against their predecessors (x264: fenc->lowres_mvs[0][0], an extra start candidate of the
 * 16x16 search in reference 0; x264_mb_predict_mv_ref16x16): device array [streams][mb_count][2] int16 in quarter-pels of the
 * half-resolution planes, first entry 0x7fff = absent for that stream.  NULL (the default) = none. */
int  x264gpu_encoder_set_lowres_mvs(x264gpu_encoder *enc, const int16_t *d_mvs);
/* the same towards the first picture of list 1 of the next B picture (x264 lowres_mvs[1][distance - 1]) */
int  x264gpu_encoder_set_lowres_mvs1(x264gpu_encoder *enc, const int16_t *d_mvs);
/* tests: the 460 CABAC context variables ((pStateIdx << 1) | valMPS) the wavefront of (stream, slice) ended the last picture with (RD sessions
 * with cabac: x264 prices candidates on the states the finished macroblocks left, [x264-upstream] encoder/rdo.c) */
int  x264gpu_encoder_cabac_states(x264gpu_encoder *enc, int stream, int slice, uint8_t *out460);

/* ---- the lookahead's frame costs in x264's structure: x264_slicetype_frame_cost(p0, p1, b) for ANY triple of pictures held in the lookahead
 * ([x264-upstream] encoder/slicetype.c slicetype_frame_cost / slicetype_slice_cost / slicetype_mb_cost, behind codec.c:1693) — I, P and B
 * costs on the half-resolution planes, blocks visited in reverse raster order with the neighbours' vectors as predictors, each picture's
 * searches cached per (list, distance) as x264's lowres_mvs / lowres_mv_costs, the intra cost computed once per picture.  What the host's
 * x264_slicetype_analyse restatement (scenecut, --b-adapt 1) and the main encoder's search candidates (x264gpu_encoder_set_lowres_mvs) read.
 * Pictures live in numbered slots; d0 = b - p0 and d1 = p1 - b are display distances (d0 = d1 = 0: the I cost; d1 = 0: a P cost). ---- */
typedef struct x264gpu_slicetype x264gpu_slicetype;
int  x264gpu_slicetype_create(x264gpu_slicetype **st, int width, int height, int streams, int slots, int bframes, int me_method, int subme, int me_range,
                              int weightb, int mv_range, int do_edges /* x264: mbtree or VBV sessions also cost the picture's edge blocks */);
void x264gpu_slicetype_destroy(x264gpu_slicetype *st);
/* x264_frame_init_lowres of a new source picture ([streams] tight I420 pictures in device memory) into `slot`; forgets the slot's cached searches */
int  x264gpu_slicetype_put_frame(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, void *stream);
/* the score of picture slot_b predicted from slot_p0 (and slot_p1): h_score[streams] (host memory; memoised per (slot_b, d0, d1)) */
int  x264gpu_slicetype_frame_cost(x264gpu_slicetype *st, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int32_t *h_score, void *stream);
/* ... of a P cost (d1 = 0) that is searched for the first time, on the reference weighted by the explicit luma weight the lookahead's
 * x264_weights_analyse found (m[0].weight in slicetype_mb_cost) */
int  x264gpu_slicetype_frame_cost_w(x264gpu_slicetype *st, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int on, int scale, int denom, int offset, int32_t *h_score, void *stream);
int  x264gpu_slicetype_intra_mbs(x264gpu_slicetype *st, int slot, int d0, int stream_idx);          /* frame->i_intra_mbs[d0] of the last P cost */
int  x264gpu_slicetype_cost_est(x264gpu_slicetype *st, int slot, int d0, int d1, int stream_idx);   /* frame->i_cost_est[d0][d1], -1 = not computed */
/* device pointers of a picture's cached results ([streams][blocks]): vectors ([2] each, quarter samples of the half-resolution planes) and costs of
 * the search in `list` at distance `dist` (NULL: not searched), per-block intra costs, lowres_costs[d0][d1] (cost | list_used << 14) */
const int16_t  *x264gpu_slicetype_lowres_mvs(x264gpu_slicetype *st, int slot, int list, int dist);
const int      *x264gpu_slicetype_lowres_mv_costs(x264gpu_slicetype *st, int slot, int list, int dist);
const int      *x264gpu_slicetype_intra_costs(x264gpu_slicetype *st, int slot);
const uint16_t *x264gpu_slicetype_lowres_costs(x264gpu_slicetype *st, int slot, int d0, int d1);
/* x264_weights_analyse's primitives (--weightp): the luma statistics of a source picture (x264_adaptive_quant_frame's i_pixel_sum / i_pixel_ssd over the
 * mod-16 expanded picture: h_out[streams][2]) and weight_cost_luma — per 8x8 block min(mbcmp(weighted reference, source), intra cost) over the whole
 * half-resolution picture + the slice-header bits of the weight; the reference motion-compensated by the picture's list-0 vectors of distance `dist`
 * when that search has run.  The picture's intra costs must exist (any x264gpu_slicetype_frame_cost of it).  The analysis around them is the caller's. */
int  x264gpu_slicetype_pixel_stats(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, uint64_t *h_out, void *stream);
int  x264gpu_slicetype_weight_cost(x264gpu_slicetype *st, int slot_fenc, int slot_ref, int dist, int on, int scale, int denom, int offset, int64_t *h_cost, void *stream);
/* ... and the chroma planes' part of x264_weights_analyse (it weights them once luma got a weight): the statistics h_out[streams][4] = { sum Cb,
 * ssd Cb, sum Cr, ssd Cr } over the mod-16 expanded picture's chroma, and weight_cost_chroma of plane 1 (Cb) / 2 (Cr) — the full-resolution chroma
 * plane of the source picture d_i420_fenc against the reference picture's d_i420_ref (both raw I420, `streams` of them), the reference
 * motion-compensated per 8x8 chroma block by the half-resolution vectors of (slot_fenc, list 0, dist) when that search has run, weighted
 * (on = 0: not); per block |sum of differences| + the slice-header bits of the weight ([x264-upstream] encoder/slicetype.c weight_cost_init_chroma,
 * weight_cost_chroma, pixel_asd8) */
int  x264gpu_slicetype_chroma_stats(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, uint64_t *h_out, void *stream);
int  x264gpu_slicetype_weight_cost_chroma(x264gpu_slicetype *st, int slot_fenc, const uint8_t *d_i420_fenc, const uint8_t *d_i420_ref, int dist, int plane,
                                          int on, int scale, int denom, int offset, int64_t *h_cost, void *stream);
/* macroblock-tree through B pictures: the building blocks of x264's macroblock_tree (the caller walks the pictures of the lookahead as x264 does:
 * clear the propagate cost of a non-B picture, x264gpu_slicetype_frame_cost of a triple, _propagate it, ..., _finish the picture about to be coded).
 * AQ offsets (x264_adaptive_quant_frame; x264gpu_lookahead_aq_offsets) weight the costs (i_inv_qscale_factor) and are the base of the result;
 * the result is what x264gpu_encoder_set_mb_qp_offsets takes.  Propagate costs saturate at 32767 as x264's do (applied where a sum is read). */
int  x264gpu_slicetype_set_aq(x264gpu_slicetype *st, int slot, const float *d_aq, void *stream);
/* --b-bias (param.i_bframe_bias, -90 .. 100): slicetype_frame_cost scales B costs by 100 / (120 + bias); call once after create */
int  x264gpu_slicetype_set_bframe_bias(x264gpu_slicetype *st, int bias);
/* how x264gpu_slicetype_frame_cost walks a picture's block rows: 0 = one wavefront per row, chained bottom-up by progress counters (default, and what -1 =
 * auto picks: measured faster at 1 and at 2048 streams), 1 = one wavefront per stream walks every row itself.  The costs are the same either way
 * (tests/test_gpu_lookahead.py). */
int  x264gpu_slicetype_set_row_mode(x264gpu_slicetype *st, int serial);
/* fenc->i_cost_est_aq[d0][d1] of a triple whose cost has been computed: the block costs weighted with the inverse quantiser scale of the picture's AQ
 * offsets (set_aq), per stream into h_score[streams] — the complexity x264_rc_analyse_slice hands the rate control in AQ sessions without macroblock-tree
 * ([x264-upstream] encoder/slicetype.c slicetype_mb_cost, ratecontrol.c x264_rc_analyse_slice) */
int  x264gpu_slicetype_cost_aq(x264gpu_slicetype *st, int slot, int d0, int d1, int32_t *h_score, void *stream);
int  x264gpu_slicetype_clear_propagate(x264gpu_slicetype *st, int slot, void *stream);
int  x264gpu_slicetype_propagate(x264gpu_slicetype *st, int slot_p0, int slot_p1, int slot_b, int d0, int d1, int referenced, void *stream);
/* macroblock_tree_finish: d_out = f_qp_offset_aq - strength * (x264_log2(intra + propagated) - x264_log2(intra) + weightdelta), strength = 5.0f * (1.0f - qcomp),
 * weightdelta = 1 - f_weighted_cost_delta[distance to reference 0 - 1] when the lookahead's weight analysis found a weight for that distance, else 0 */
int  x264gpu_slicetype_finish(x264gpu_slicetype *st, int slot, float strength, float weightdelta, float *d_out, void *stream);
const int32_t  *x264gpu_slicetype_propagate_cost(x264gpu_slicetype *st, int slot);

/* ------------------------------------------------------------------------------------------------
 * Lookahead frame cost: x264_slicetype_frame_cost of [x264-upstream]
 * encoder/slicetype.c for the pair p0 = previous picture, b = p1 = new picture, on half-resolution planes
 * (frame_init_lowres + 8x8 hexagon / sub-pel search + 8x8 intra SATD, lambda of qp 12).  Feeds the scenecut decision and
 * the CRF complexity (x264_rc_analyse_slice) of the host encoder; independent of the encoder's DPB (source pictures only).
 * d_i420: `streams` tightly packed I420 pictures.  reset != 0 forgets the previous picture (first picture / after an IDR
 * decided elsewhere is NOT a reset: x264 keeps comparing consecutive source pictures).
 * d_out [streams][4] int32: intra cost (i_cost_est[0][0]), P cost (i_cost_est[1][0]; = intra cost without a previous
 * picture), blocks where intra won, blocks in the frame score.  d_blocks (optional) [streams][blocks][4]: intra cost, best cost, the
 * block's vector (x & 0xffff | y << 16, quarter-pel of the half-resolution planes; 0 when intra won), 1 if inter won — the record
 * x264gpu_lookahead_mbtree consumes.
 * ------------------------------------------------------------------------------------------------ */
typedef struct x264gpu_lookahead x264gpu_lookahead;
int  x264gpu_lookahead_create(x264gpu_lookahead **la, int width, int height, int streams, int me_range, int subme);
void x264gpu_lookahead_destroy(x264gpu_lookahead *la);
int  x264gpu_lookahead_frame_cost(x264gpu_lookahead *la, const uint8_t *d_i420, int reset, int32_t *d_out, int32_t *d_blocks, void *stream);
/* Adaptive-quantisation offsets (single floats) of `streams` source pictures: strength * (x264_log2(AC energy) - 14.427f) per macroblock, as the lookahead
 * of x264 computes them when a picture arrives (x264_adaptive_quant_frame); what aq_mode 1 of the encoder computes itself.  strength = aq-strength * 1.0397f */
int  x264gpu_lookahead_aq_offsets(x264gpu_lookahead *la, const uint8_t *d_i420, float strength, float *d_out, void *stream);
/* ... with x264's --aq-mode: 1 = the call above; 2 (auto-variance) / 3 (auto-variance with a bias to dark scenes): x264_adaptive_quant_frame's float
 * path — per macroblock (energy + 1)^(1/8) (three IEEE square roots here; x264 calls powf, which may differ in the last place), the picture's mean and mean
 * square of them (summed in raster order), strength x (value - average) [+ aq-strength x (1 - 14 / value^2)]; strength = the plain aq-strength for modes 2 / 3
 * (mode 1: x 1.0397f as above) */
int  x264gpu_lookahead_aq_offsets_mode(x264gpu_lookahead *la, const uint8_t *d_i420, int mode, float strength, float *d_out, void *stream);
/* Macroblock-tree ([x264-upstream] encoder/slicetype.c macroblock_tree / _propagate / _finish, common/mc.c mbtree_propagate_cost / _list)
 * for an I/P-only stream at constant frame rate.  d_info[j], d_aq[j] (host arrays of n device pointers): the per-block records
 * (x264gpu_lookahead_frame_cost d_blocks) and AQ offsets of n consecutive pictures, j = 0 the one about to be coded; d_aq may be NULL.
 * d_out [streams][blocks]: quantiser offsets of picture 0 = aq - strength * (x264_log2(intra + propagated) - x264_log2(intra));
 * strength = 5.0f * (1.0f - qcomp).  Feed it to x264gpu_encoder_set_mb_qp_offsets.  x264's C expressions in single floats throughout. */
int  x264gpu_lookahead_mbtree(x264gpu_lookahead *la, const int32_t *const *d_info, const float *const *d_aq, int n, float strength,
                              float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif

/* oracle/x264o.h — CPU restatement of the H.264 encode hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The product path (x264vfw_amd/) never links it.
 *
 * PARITY STATUS: "parity unpinned" against real libx264.  The reference tree (/root/reference) is the
 * x264vfw driver shell; every function restated here lives in libx264, an un-vendored, un-pinned
 * link-time dependency (reference Makefile:21-23,103,109; common.h:39-40; inferred X264_BUILD 155-159).
 * The only call site of this path in the reference is x264_encoder_encode() at codec.c:1693.
 * What IS pinned: every normative H.264 operation here (inverse transforms, dequant, intra predictors,
 * 6-tap/qpel/chroma interpolation, deblocking) is checked in tests/ against an independent numpy
 * restatement of ITU-T H.264 clauses 8.3-8.7 (tests/spec_ref.py), and the bitstream is closed-loop
 * checked by the decoder in oracle/h264dec.cpp; parameter sets and slice headers are additionally read back by L-SMASH's
 * parser from the reference tree (oracle/_ref, built by the Makefile from /root/reference/output/L-SMASH + lsmash_shim.c).
 * Non-normative arithmetic (SAD/SATD/SA8D, forward
 * transforms, deadzone quant, decimation, search order, cost tables) restates x264's published
 * algorithm from its public description.
 */
#ifndef X264O_H
#define X264O_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t pixel;
typedef int16_t dctcoef;

/* ---- pixel metrics: [x264-upstream] common/pixel.c (A2/A3) ---- */
int      x264o_sad(const pixel *a, int sa, const pixel *b, int sb, int w, int h);
int      x264o_ssd(const pixel *a, int sa, const pixel *b, int sb, int w, int h);
int      x264o_satd(const pixel *a, int sa, const pixel *b, int sb, int w, int h);
int      x264o_sa8d_8x8_raw(const pixel *a, int sa, const pixel *b, int sb); /* un-normalised sum */
int      x264o_sa8d(const pixel *a, int sa, const pixel *b, int sb, int w, int h); /* 8x8 or 16x16 */
uint64_t x264o_var(const pixel *p, int stride, int w, int h); /* sum | (sqr << 32) */
uint64_t x264o_hadamard_ac(const pixel *p, int stride, int w, int h); /* (sum8<<32)|sum4, psy-RD */

/* ---- transforms: common/dct.c (A6/A8).  Blocks are row-major: c[row*N+col] ---- */
void x264o_sub4x4_dct(dctcoef d[16], const pixel *enc, int se, const pixel *pred, int sp);
void x264o_add4x4_idct(pixel *dst, int sd, const dctcoef d[16]);
void x264o_sub8x8_dct8(dctcoef d[64], const pixel *enc, int se, const pixel *pred, int sp);
void x264o_add8x8_idct8(pixel *dst, int sd, const dctcoef d[64]);
void x264o_dct4x4dc(dctcoef d[16]);   /* forward luma-DC Hadamard, (x+1)>>1 */
void x264o_idct4x4dc(dctcoef d[16]);  /* inverse luma-DC Hadamard (normative 8.5.10) */
void x264o_dct2x2dc(dctcoef d[4]);    /* chroma DC Hadamard (self-inverse up to scale) */
void x264o_add4x4_idct_dc(pixel *dst, int sd, int dc); /* DC-only shortcut: (dc+32)>>6 added */
extern const uint8_t x264o_zigzag4[16];  /* scan idx -> raster pos (frame) */
extern const uint8_t x264o_zigzag8[64];

/* ---- quant: common/quant.c + common/set.c cqm init (A7).  Flat CQM only ---- */
enum { X264O_CQM_4IY = 0, X264O_CQM_4PY = 1, X264O_CQM_4IC = 2, X264O_CQM_4PC = 3,
       X264O_CQM_8IY = 0, X264O_CQM_8PY = 1 };
typedef struct {
    uint16_t quant4_mf[4][52][16];
    uint16_t quant4_bias[4][52][16];
    int32_t  dequant4_mf[6][16];
    uint16_t quant8_mf[2][52][64];
    uint16_t quant8_bias[2][52][64];
    int32_t  dequant8_mf[6][64];
} x264o_quant_tables;
/* deadzone_inter / deadzone_intra: --deadzone-inter 21 / --deadzone-intra 11 (config.c:1688-1689) */
void x264o_quant_init(x264o_quant_tables *t, int deadzone_inter, int deadzone_intra);
int  x264o_quant_4x4(dctcoef d[16], const uint16_t mf[16], const uint16_t bias[16]);
int  x264o_quant_8x8(dctcoef d[64], const uint16_t mf[64], const uint16_t bias[64]);
int  x264o_quant_4x4_dc(dctcoef d[16], int mf, int bias);
int  x264o_quant_2x2_dc(dctcoef d[4], int mf, int bias);
void x264o_dequant_4x4(dctcoef d[16], const int32_t dq[6][16], int qp);
void x264o_dequant_8x8(dctcoef d[64], const int32_t dq[6][64], int qp);
void x264o_dequant_4x4_dc(dctcoef d[16], const int32_t dq[6][16], int qp);
void x264o_dequant_2x2_dc(dctcoef out[4], const dctcoef in[4], const int32_t dq[6][16], int qp);
int  x264o_optimize_chroma_2x2_dc(dctcoef d[4], int dmf);   /* dmf = dequant4_mf[qp%6][0] << qp/6; 0 = nothing left to code */
int  x264o_decimate_score(const dctcoef *scanned, int n); /* n = 15, 16 or 64; scanned order */
int  x264o_coeff_last(const dctcoef *scanned, int n);
extern const uint8_t x264o_chroma_qp[52];

/* ---- intra prediction: common/predict.c (A5), normative 8.3 ---- */
enum { I_PRED_16x16_V = 0, I_PRED_16x16_H, I_PRED_16x16_DC, I_PRED_16x16_P,
       I_PRED_16x16_DC_LEFT, I_PRED_16x16_DC_TOP, I_PRED_16x16_DC_128 };
enum { I_PRED_CHROMA_DC = 0, I_PRED_CHROMA_H, I_PRED_CHROMA_V, I_PRED_CHROMA_P,
       I_PRED_CHROMA_DC_LEFT, I_PRED_CHROMA_DC_TOP, I_PRED_CHROMA_DC_128 };
enum { I_PRED_4x4_V = 0, I_PRED_4x4_H, I_PRED_4x4_DC, I_PRED_4x4_DDL, I_PRED_4x4_DDR,
       I_PRED_4x4_VR, I_PRED_4x4_HD, I_PRED_4x4_VL, I_PRED_4x4_HU,
       I_PRED_4x4_DC_LEFT, I_PRED_4x4_DC_TOP, I_PRED_4x4_DC_128 };
/* neighbour availability bit-set */
enum { X264O_AVAIL_LEFT = 1, X264O_AVAIL_TOP = 2, X264O_AVAIL_TOPRIGHT = 4, X264O_AVAIL_TOPLEFT = 8 };
/* All predictors read neighbours from `src` (pointer to the block's top-left sample inside a
 * reconstructed plane, stride ss) and write the w x h prediction to dst (stride sd). */
void x264o_predict_16x16(pixel *dst, int sd, const pixel *src, int ss, int mode);
void x264o_predict_8x8c(pixel *dst, int sd, const pixel *src, int ss, int mode);
void x264o_predict_4x4(pixel *dst, int sd, const pixel *src, int ss, int mode, int avail);
/* 8x8: edge[33] layout produced by predict_8x8_filter: [0..7]=left bottom..top (l7..l0),
 * [8]=topleft, [9..16]=top, [17..24]=topright (after the 1-2-1 filter of 8.3.2.2.1) */
void x264o_predict_8x8_filter(const pixel *src, int ss, pixel edge[33], int avail);
void x264o_predict_8x8(pixel *dst, int sd, const pixel edge[33], int mode);

/* ---- lookahead frame cost: encoder/slicetype.c (A12, next-row f2) — oracle/lookahead.c ---- */
typedef struct x264o_lookahead x264o_lookahead;
x264o_lookahead *x264o_lookahead_create(int width, int height, int me_range, int subme);
void x264o_lookahead_destroy(x264o_lookahead *la);
/* out[4] = { intra cost, P cost vs the previous picture, intra blocks, blocks in the frame score }; block_info (blocks x 4 int32) optional */
int  x264o_lookahead_frame_cost(x264o_lookahead *la, const uint8_t *i420, int reset, int32_t out[4], int32_t *block_info);
/* per-macroblock AQ offsets (single floats) of a source picture; macroblock-tree offsets of picture 0 from n consecutive pictures' records */
void x264o_aq_offsets(const uint8_t *i420, int w, int h, float strength, float *out);
void x264o_aq_offsets_mode(const uint8_t *i420, int w, int h, int mode, float strength, float *out);
void x264o_mbtree(int bw, int bh, const int32_t *const *info, const float *const *aq, int n, float strength, float *out);

/* ---- motion compensation / plane filters: common/mc.c, common/frame.c (A1/A4/A10) ---- */
/* hpel planes: dsth/dstv/dstc (H, V, centre=HV) for an w x h region; src is read with
 * coordinates clamped to [0,w-1]x[0,h-1] by the caller-supplied clamp dims (planes are produced
 * for the padded area directly, see x264o_frame_filter). */
void x264o_hpel_filter_pixel(const pixel *src, int stride, int w, int h, int x, int y,
                             pixel *ph, pixel *pv, pixel *pc);
/* plane[0..3] = full, H, V, HV; each padded by `pad` on every side, stride `stride`;
 * pointers address sample (0,0). Fills all four planes over [-pad, w+pad) x [-pad, h+pad). */
void x264o_frame_filter(pixel *plane[4], int stride, int w, int h, int pad);
/* lowres: four half-resolution planes from luma (clamped reads); dst stride ds, size (w/2)x(h/2) */
void x264o_frame_init_lowres(const pixel *src, int ss, int w, int h, pixel *dst[4], int ds);
/* qpel luma fetch (normative 8.4.2.2.1 via hpel planes).  mv in quarter-pels relative to (0,0) of plane */
void x264o_pixel_avg_weight(pixel *dst, int sd, const pixel *a, int sa, const pixel *b, int sb, int w, int h, int weight1);
void x264o_mc_weight(pixel *dst, int sd, const pixel *src, int ss, int w, int h, int scale, int denom, int offset);
void x264o_mc_luma(pixel *dst, int sd, pixel *const plane[4], int stride, int x, int y,
                   int mvx, int mvy, int w, int h);
/* chroma 1/8-pel bilinear on an NV12 plane (normative 8.4.2.2.2): dstu/dstv w x h (chroma samples) */
void x264o_mc_chroma(pixel *dstu, pixel *dstv, int sd, const pixel *nv12, int stride,
                     int x, int y, int mvx, int mvy, int w, int h);
void x264o_pixel_avg(pixel *dst, int sd, const pixel *a, int sa, const pixel *b, int sb, int w, int h);

/* ---- deblocking: common/deblock.c (A9), normative 8.7 ---- */
extern const uint8_t x264o_alpha_table[52];
extern const uint8_t x264o_beta_table[52];
extern const uint8_t x264o_tc0_table[52][3];
/* filter one luma edge segment of 4 lines; xstride = step across the edge, ystride = along it */
void x264o_deblock_luma_edge(pixel *pix, int xstride, int ystride, int lines, int alpha, int beta, int tc0, int bs);
void x264o_deblock_chroma_edge(pixel *pix, int xstride, int ystride, int lines, int alpha, int beta, int tc0, int bs);

/* ---- input colourspace conversion to I420: /root/reference/csp.c (next-row f1) ---- */
enum { X264O_CSP_MASK = 0xff, X264O_CSP_I420 = 1, X264O_CSP_YV12 = 2, X264O_CSP_YV16 = 3, X264O_CSP_YV24 = 4, X264O_CSP_NV12 = 5,
       X264O_CSP_YUYV = 6, X264O_CSP_UYVY = 7, X264O_CSP_BGR = 8, X264O_CSP_BGRA = 9, X264O_CSP_VFLIP = 0x1000 };   /* csp.h:30-44 */
void x264o_csp_rgb_coefs(int colmatrix709, int fullrange, uint32_t c[12]);
long x264o_csp_img_fill(int csp, int width, int height, long off[3], int stride[3]);
int  x264o_csp_to_i420(uint8_t *const dst[3], const int dstride[3], const uint8_t *const src[3], const int sstride[3],
                       int csp, int w, int h, int colmatrix709, int fullrange);

#ifdef __cplusplus
}
#endif
#endif

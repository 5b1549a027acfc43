/* oracle/encoder.c — CPU restatement of the per-frame encode hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the executable specification of the frame pipeline behind x264_encoder_encode()
 * (reference call site codec.c:1693; [x264-upstream] encoder/{analyse,me,macroblock}.c,
 * common/{deblock,frame,mc}.c).  The HIP pipeline (x264vfw_amd/csrc/encoder.hip) must reproduce its
 * x264gpu_mb records, quantised levels and reconstructed frames bit-exactly.
 *
 * parity unpinned vs libx264 (not in /root/reference, see x264o.h).  Deliberate structural choices that
 * differ from x264's raster-serial macroblock loop, made so that every stage is data-parallel on the
 * GPU and still a conformant H.264 encoder (documented in DESIGN.md "pipeline"):
 *   - P-frame ME predicts MVs from the PREVIOUS frame's MV field (median of left/top/topright) instead
 *     of the current frame's already-coded neighbours; the bitstream mvd is still derived from the
 *     true H.264 predictor at entropy time on the host.
 *   - the intra/inter decision in P frames uses an intra-16x16 SATD estimate on source neighbours;
 *     intra macroblocks are then analysed and coded with real reconstructed neighbours.
 *   - P_Skip is detected at entropy time (16x16, ref 0, mv == skip predictor, no residual).
 * Primitive arithmetic (SAD/SATD, hex + subpel search order and tie-breaks, transforms, deadzone
 * quant, decimation, intra costs) follows x264 as described in SURVEY.md Appendix C.
 */
#include "x264o.h"
#include "x264gpu.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PAD 32          /* luma padding of reference planes */
#define CPAD 16         /* chroma padding (samples) */
#define MVCOST_HALF 32768
#define X264O_MAX_SLOTS 5     /* up to 4 reference frames + current */

typedef struct x264o_encoder {
    x264gpu_config cfg;
    int mbw, mbh, cw, ch;
    int fs;                      /* fenc stride (luma and NV12 chroma) */
    pixel *fenc_y, *fenc_uv;
    int rs;                      /* reference plane stride */
    size_t plane_bytes, cplane_bytes;
    pixel *luma[X264O_MAX_SLOTS];   /* DPB slots: 4 padded planes each (refs + the picture being built) */
    pixel *chroma[X264O_MAX_SLOTS]; /* padded NV12 */
    int slots;                   /* refs + 1 */
    int nref;                    /* reference pictures usable by the current P slice */
    int cur;                     /* DPB slot being reconstructed */
    int16_t (*mvf[2])[2];        /* per-MB mv field: [0] previous frame, [1] current */
    int8_t *reff[2];             /* per-MB ref (-1 = intra) */
    uint16_t *cost_mv[52];       /* lambda-scaled mv bit costs per qp, centred at MVCOST_HALF */
    x264o_quant_tables qt;
    int have_ref;
    int slice_type;              /* slice being encoded */
    uint8_t *mbqp;               /* quantiser of every macroblock of the picture being coded (slice quantiser, + AQ offset) */
    const int16_t *ext_off_q8;   /* quantiser offsets handed in for the next picture (lookahead: AQ - macroblock-tree), or NULL */
} x264o_encoder;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static int median3(int a, int b, int c) { int mn = a < b ? a : b, mx = a < b ? b : a; return c < mn ? mn : c > mx ? mx : c; }
static int bs_size_ue(int v) { int n = 0; v++; while (v >> (n + 1)) n++; return 2 * n + 1; }

int x264o_lambda(int qp)
{
    double v = pow(2.0, qp / 6.0 - 2.0);
    int l = (int)(v + 0.5);
    return l < 1 ? 1 : l;
}

/* lambda * (2*log2(|mvd|+1) + 0.718 + (mvd != 0)) + 0.5, saturated to u16 (float math as in x264) */
void x264o_build_cost_mv(uint16_t *tab /* 2*MVCOST_HALF entries */, int lambda)
{
    for (int i = 0; i < MVCOST_HALF; i++) {
        float bits = log2f((float)(i + 1)) * 2.0f + 0.718f + (i ? 1.0f : 0.0f);
        int c = (int)((float)lambda * bits + 0.5f);
        if (c > 65535) c = 65535;
        tab[MVCOST_HALF + i] = (uint16_t)c;
        tab[MVCOST_HALF - i] = (uint16_t)c;
    }
    tab[0] = tab[1];
}

/* DPB slot of reference index r of the current P slice: r = 0 is the most recent picture */
static int ref_slot(const x264o_encoder *e, int r) { return (e->cur - 1 - r + 2 * e->slots) % e->slots; }
static pixel *luma_plane(x264o_encoder *e, int slot, int k) { return e->luma[slot] + k * e->plane_bytes + (size_t)PAD * e->rs + PAD; }
static pixel *chroma_plane(x264o_encoder *e, int slot) { return e->chroma[slot] + (size_t)CPAD * e->rs + 2 * CPAD; }

x264o_encoder *x264o_encoder_create(const x264gpu_config *cfg)
{
    x264o_encoder *e = calloc(1, sizeof(*e));
    e->cfg = *cfg;
    e->mbw = (cfg->width + 15) / 16; e->mbh = (cfg->height + 15) / 16;
    e->cw = e->mbw * 16; e->ch = e->mbh * 16;
    e->fs = (e->cw + 63) / 64 * 64;
    e->fenc_y = malloc((size_t)e->fs * e->ch);
    e->fenc_uv = malloc((size_t)e->fs * e->ch / 2);
    e->rs = (e->cw + 2 * PAD + 63) / 64 * 64;
    e->plane_bytes = (size_t)e->rs * (e->ch + 2 * PAD);
    e->cplane_bytes = (size_t)e->rs * (e->ch / 2 + 2 * CPAD);
    e->slots = clampi(cfg->refs, 1, X264O_MAX_SLOTS - 1) + 1;
    for (int s = 0; s < e->slots; s++) {
        e->luma[s] = calloc(4, e->plane_bytes);
        e->chroma[s] = calloc(1, e->cplane_bytes);
    }
    for (int s = 0; s < 2; s++) {
        e->mvf[s] = calloc((size_t)e->mbw * e->mbh, sizeof(int16_t[2]));
        e->reff[s] = malloc((size_t)e->mbw * e->mbh);
        memset(e->reff[s], -1, (size_t)e->mbw * e->mbh);
    }
    e->mbqp = malloc((size_t)e->mbw * e->mbh);
    x264o_quant_init(&e->qt, cfg->deadzone_inter, cfg->deadzone_intra);
    return e;
}

void x264o_encoder_destroy(x264o_encoder *e)
{
    if (!e) return;
    for (int s = 0; s < e->slots; s++) { free(e->luma[s]); free(e->chroma[s]); }
    for (int s = 0; s < 2; s++) { free(e->mvf[s]); free(e->reff[s]); }
    for (int q = 0; q < 52; q++) free(e->cost_mv[q]);
    free(e->fenc_y); free(e->fenc_uv); free(e->mbqp); free(e);
}

int x264o_encoder_mb_count(const x264o_encoder *e) { return e->mbw * e->mbh; }
void x264o_encoder_set_qp(x264o_encoder *e, int qp_i, int qp_p) { e->cfg.qp_i = qp_i; e->cfg.qp_p = qp_p; }
/* per-macroblock quantiser offsets (Q8) for the following pictures; the array must stay valid; NULL = back to the encoder's own AQ */
void x264o_encoder_set_mb_qp_offsets(x264o_encoder *e, const int16_t *off_q8) { e->ext_off_q8 = off_q8; }

static const uint16_t *cost_mv_for(x264o_encoder *e, int qp)
{
    if (!e->cost_mv[qp]) {
        e->cost_mv[qp] = malloc(2 * MVCOST_HALF * sizeof(uint16_t));
        x264o_build_cost_mv(e->cost_mv[qp], x264o_lambda(qp));
    }
    return e->cost_mv[qp] + MVCOST_HALF;
}

/* ---- stage 0: ingest (x264_frame_copy_picture + expand_border_mod16; A1) ---- */
static void ingest(x264o_encoder *e, const uint8_t *i420)
{
    int w = e->cfg.width, h = e->cfg.height;
    const uint8_t *sy = i420, *su = i420 + (size_t)w * h, *sv = su + (size_t)(w / 2) * (h / 2);
    for (int y = 0; y < e->ch; y++)
        for (int x = 0; x < e->cw; x++)
            e->fenc_y[(size_t)y * e->fs + x] = sy[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
    for (int y = 0; y < e->ch / 2; y++)
        for (int x = 0; x < e->cw / 2; x++) {
            size_t so = (size_t)clampi(y, 0, h / 2 - 1) * (w / 2) + clampi(x, 0, w / 2 - 1);
            e->fenc_uv[(size_t)y * e->fs + 2 * x] = su[so];
            e->fenc_uv[(size_t)y * e->fs + 2 * x + 1] = sv[so];
        }
}

/* ---- residual coding helpers ---- */
/* ---- adaptive quantisation, mode 1 (x264_adaptive_quant_frame / x264_ac_energy_mb, [x264-upstream] encoder/ratecontrol.c):
 * energy = var(16x16 luma) + var(8x8 U) + var(8x8 V) with var = ssd - (sum^2 >> log2 n); offset = strength * (log2(energy) - 14.427).
 * Q8 fixed point: log2 = 256 * floor(log2 e) + table[next 7 bits]; the offset is rounded to an integer quantiser step. ---- */
static const uint8_t aq_log2_lut[128] = {
#include "x264gpu_aq_lut.inc"
};
static int aq_log2_q8(uint32_t x)
{
    int lz = 31 - __builtin_clz(x);
    return lz * 256 + aq_log2_lut[((x << (31 - lz)) >> 24) & 0x7f];
}
static void compute_mb_qp(x264o_encoder *e, int slice_qp)
{
    const int n = e->mbw * e->mbh;
    if (e->ext_off_q8) {        /* offsets decided by the lookahead (x264: frame->f_qp_offset, read by x264_ratecontrol_mb_qp) */
        for (int i = 0; i < n; i++) e->mbqp[i] = (uint8_t)clampi(slice_qp + ((e->ext_off_q8[i] + 128) >> 8), 1, 51);
        return;
    }
    if (!e->cfg.aq_mode) { memset(e->mbqp, slice_qp, (size_t)n); return; }
    for (int mby = 0; mby < e->mbh; mby++)
        for (int mbx = 0; mbx < e->mbw; mbx++) {
            const pixel *y = e->fenc_y + (size_t)mby * 16 * e->fs + mbx * 16, *uv = e->fenc_uv + (size_t)mby * 8 * e->fs + mbx * 16;
            uint32_t sum = 0, sqr = 0, su = 0, squ = 0, sv = 0, sqv = 0;
            for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { uint32_t p = y[r * e->fs + c]; sum += p; sqr += p * p; }
            for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) {
                uint32_t u = uv[r * e->fs + 2 * c], v = uv[r * e->fs + 2 * c + 1];
                su += u; squ += u * u; sv += v; sqv += v * v;
            }
            uint32_t energy = (sqr - (sum * sum >> 8)) + (squ - (su * su >> 6)) + (sqv - (sv * sv >> 6));
            int adj_q8 = (e->cfg.aq_strength_q8 * (aq_log2_q8(energy ? energy : 1) - 3693)) >> 8;      /* 14.427 * 256 = 3693 */
            e->mbqp[mby * e->mbw + mbx] = (uint8_t)clampi(slice_qp + ((adj_q8 + 128) >> 8), 1, 51);
        }
}

/* QP_Y of a macroblock that sends no mb_qp_delta (no coefficients, not Intra16x16) is the previous macroblock's (7.4.5): the
 * deblocking filter and the next delta use that value (x264_macroblock_cache_save does the same) */
static void settle_mb_qp(x264o_encoder *e, x264gpu_mb *mbs, int slice_qp)
{
    int last = slice_qp;
    for (int i = 0; i < e->mbw * e->mbh; i++) {
        x264gpu_mb *m = &mbs[i];
        if (m->type != X264GPU_MB_I16x16 && !m->cbp_luma && !m->cbp_chroma) m->qp = (uint8_t)last;
        last = m->qp;
    }
}

static void scan4(int16_t *dst, const dctcoef *src) { for (int k = 0; k < 16; k++) dst[k] = src[x264o_zigzag4[k]]; }

static const uint8_t blk_x[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
static const uint8_t blk_y[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };

/* inter luma: fenc vs prediction already in rec (16x16 at rec, stride rs); writes recon in place */
static void encode_luma_inter(x264o_encoder *e, const pixel *fenc, pixel *rec, int qp, x264gpu_mb *mb, int16_t *lv)
{
    dctcoef d[16][16];
    int nz[16], score8[4] = { 0, 0, 0, 0 };
    const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4PY][qp], *bias = e->qt.quant4_bias[X264O_CQM_4PY][qp];
    for (int b = 0; b < 16; b++) {
        const pixel *f = fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4;
        pixel *r = rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4;
        x264o_sub4x4_dct(d[b], f, e->fs, r, e->rs);
        nz[b] = x264o_quant_4x4(d[b], mf, bias);
        scan4(lv + b * 16, d[b]);
        if (nz[b] && e->cfg.dct_decimate) score8[b >> 2] += x264o_decimate_score(lv + b * 16, 16);
    }
    int mbscore = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        int any = nz[i8 * 4] | nz[i8 * 4 + 1] | nz[i8 * 4 + 2] | nz[i8 * 4 + 3];
        if (any) mbscore += score8[i8];               /* every coded 8x8 counts towards the macroblock score, kept or not */
        if (any && e->cfg.dct_decimate && score8[i8] < 4) any = 0;
        if (!any) for (int k = 0; k < 4; k++) nz[i8 * 4 + k] = 0;
    }
    if (e->cfg.dct_decimate && mbscore < 6) for (int b = 0; b < 16; b++) nz[b] = 0;
    for (int b = 0; b < 16; b++) {
        if (!nz[b]) { memset(lv + b * 16, 0, 32); continue; }
        x264o_dequant_4x4(d[b], e->qt.dequant4_mf, qp);
        x264o_add4x4_idct(rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs, d[b]);
        mb->nnz |= 1u << b;
        mb->cbp_luma |= 1 << (b >> 2);
    }
}

/* inter luma with the 8x8 transform ([x264-upstream] encoder/macroblock.c x264_macroblock_encode, b_transform_8x8
 * branch): sub16x16_dct8, quant_8x8, scan_8x8, decimate_score64 per 8x8 (kept when >= 4) and per MB (>= 6),
 * dequant_8x8 + add8x8_idct8; levels leave in the CAVLC-interleaved 4x4 form (zigzag_interleave_8x8_cavlc). */
static void encode_luma_inter8(x264o_encoder *e, const pixel *fenc, pixel *rec, int qp, x264gpu_mb *mb, int16_t *lv)
{
    dctcoef d[4][64];
    int16_t scan[4][64];
    int keep[4], mbscore = 0;
    const uint16_t *mf = e->qt.quant8_mf[X264O_CQM_8PY][qp], *bias = e->qt.quant8_bias[X264O_CQM_8PY][qp];
    for (int i8 = 0; i8 < 4; i8++) {
        const pixel *f = fenc + (i8 >> 1) * 8 * e->fs + (i8 & 1) * 8;
        pixel *r = rec + (i8 >> 1) * 8 * e->rs + (i8 & 1) * 8;
        x264o_sub8x8_dct8(d[i8], f, e->fs, r, e->rs);
        keep[i8] = x264o_quant_8x8(d[i8], mf, bias);
        for (int k = 0; k < 64; k++) scan[i8][k] = d[i8][x264o_zigzag8[k]];
        if (keep[i8] && e->cfg.dct_decimate) {
            int sc = x264o_decimate_score(scan[i8], 64);
            mbscore += sc;
            if (sc < 4) keep[i8] = 0;
        }
    }
    if (e->cfg.dct_decimate && mbscore < 6) keep[0] = keep[1] = keep[2] = keep[3] = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        if (!keep[i8]) continue;            /* levels were zeroed by the caller */
        for (int k = 0; k < 64; k++) {
            int16_t v = scan[i8][k];
            lv[(i8 * 4 + (k & 3)) * 16 + (k >> 2)] = v;
            if (v) mb->nnz |= 1u << (i8 * 4 + (k & 3));
        }
        x264o_dequant_8x8(d[i8], e->qt.dequant8_mf, qp);
        x264o_add8x8_idct8(rec + (i8 >> 1) * 8 * e->rs + (i8 & 1) * 8, e->rs, d[i8]);
        mb->cbp_luma |= 1 << i8;
    }
}

/* chroma of one MB: pred already in ru/rv-interleaved NV12 recon (rec points at U of the 8x8) */
static void encode_chroma(x264o_encoder *e, const pixel *fenc_uv, pixel *rec_uv, int qpc, int inter, x264gpu_mb *mb, int16_t *lv)
{
    int list = inter ? X264O_CQM_4PC : X264O_CQM_4IC;
    const uint16_t *mf = e->qt.quant4_mf[list][qpc], *bias = e->qt.quant4_bias[list][qpc];
    int any_ac = 0, any_dc = 0;
    for (int c = 0; c < 2; c++) {
        pixel f[64], p[64];
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) { f[y * 8 + x] = fenc_uv[y * e->fs + 2 * x + c]; p[y * 8 + x] = rec_uv[y * e->rs + 2 * x + c]; }
        dctcoef d[4][16], dc[4];
        int nz[4], score = 0, nzac = 0;
        for (int i = 0; i < 4; i++) {
            int o = (i >> 1) * 32 + (i & 1) * 4;
            x264o_sub4x4_dct(d[i], f + o, 8, p + o, 8);
            dc[i] = d[i][0]; d[i][0] = 0;
            nz[i] = x264o_quant_4x4(d[i], mf, bias);
            int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
            scan4(l, d[i]);
            if (nz[i]) { nzac = 1; if (inter && e->cfg.dct_decimate) score += x264o_decimate_score(l + 1, 15); }
        }
        if (nzac && inter && e->cfg.dct_decimate && score < 7) nzac = 0;
        x264o_dct2x2dc(dc);
        int nzdc = x264o_quant_2x2_dc(dc, mf[0] >> 1, bias[0] << 1);
        /* DC-only planes (no AC left): x264_mb_optimize_chroma_dc trims the DC levels that do not change the reconstruction */
        if (nzdc && !nzac && !x264o_optimize_chroma_2x2_dc(dc, e->qt.dequant4_mf[qpc % 6][0] << (qpc / 6))) { nzdc = 0; dc[0] = dc[1] = dc[2] = dc[3] = 0; }
        for (int i = 0; i < 4; i++) lv[X264GPU_LV_CHROMA_DC + c * 4 + i] = dc[i];
        dctcoef dq[4] = { 0, 0, 0, 0 };
        if (nzdc) { x264o_dequant_2x2_dc(dq, dc, e->qt.dequant4_mf, qpc); mb->nnz |= 1u << (25 + c); any_dc = 1; }
        for (int i = 0; i < 4; i++) {
            int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
            if (!nzac || !nz[i]) { memset(l, 0, 32); memset(d[i], 0, sizeof(d[i])); }
            else { x264o_dequant_4x4(d[i], e->qt.dequant4_mf, qpc); mb->nnz |= 1u << (16 + c * 4 + i); any_ac = 1; }
            d[i][0] = dq[i];
            int o = (i >> 1) * 32 + (i & 1) * 4;
            x264o_add4x4_idct(p + o, 8, d[i]);
        }
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) rec_uv[y * e->rs + 2 * x + c] = p[y * 8 + x];
    }
    mb->cbp_chroma = any_ac ? 2 : any_dc ? 1 : 0;
}

/* ---- stage 1: P-frame analysis of one macroblock (x264_me_search_ref + refine_subpel, 16x16) ---- */
typedef struct { int mvx, mvy, cost; } me_result;

static void mv_limits(const x264o_encoder *e, int mbx, int mby, int spel_min[2], int spel_max[2], int fpel_min[2], int fpel_max[2])
{
    const int vrange = 512 * 4;   /* --mvrange default: +-512 luma rows */
    spel_min[0] = 4 * (-16 * mbx - 24); spel_max[0] = 4 * (16 * (e->mbw - mbx - 1) + 24);
    spel_min[1] = 4 * (-16 * mby - 24); spel_max[1] = 4 * (16 * (e->mbh - mby - 1) + 24);
    spel_min[1] = clampi(spel_min[1], -vrange, vrange - 1);
    spel_max[1] = clampi(spel_max[1], -vrange, vrange - 1);
    for (int k = 0; k < 2; k++) { fpel_min[k] = (spel_min[k] >> 2) + 6; fpel_max[k] = (spel_max[k] >> 2) - 6; }
}

static void prev_mvp(const x264o_encoder *e, int mbx, int mby, int mvp[2])
{
    /* median of the previous frame's left / top / topright MVs; absent or intra neighbours count as 0;
     * when neither top nor topright exists the left neighbour is used alone (H.264 8.4.1.3 flavour) */
    int16_t (*f)[2] = e->mvf[0];
    const int8_t *r = e->reff[0];
    int a[2] = { 0, 0 }, b[2] = { 0, 0 }, c[2] = { 0, 0 };
    int ia = mbx > 0, ib = mby > 0, ic = mby > 0 && mbx + 1 < e->mbw;
    if (ia) { int i = mby * e->mbw + mbx - 1; if (r[i] >= 0) { a[0] = f[i][0]; a[1] = f[i][1]; } }
    if (ib) { int i = (mby - 1) * e->mbw + mbx; if (r[i] >= 0) { b[0] = f[i][0]; b[1] = f[i][1]; } }
    if (ic) { int i = (mby - 1) * e->mbw + mbx + 1; if (r[i] >= 0) { c[0] = f[i][0]; c[1] = f[i][1]; } }
    else if (mby > 0 && mbx > 0) { int i = (mby - 1) * e->mbw + mbx - 1; if (r[i] >= 0) { c[0] = f[i][0]; c[1] = f[i][1]; } }
    if (!ib && ia) { mvp[0] = a[0]; mvp[1] = a[1]; }
    else { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
}

static const int8_t hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
static const int8_t square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
static const int8_t mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };

/* SATD of the two chroma planes of a w x h LUMA block at luma offset (ox,oy), predicted with quarter-pel vector (mvx,mvy):
 * the chroma term of [x264-upstream] me.c COST_MV_SATD under b_chroma_me (mc_chroma + mbcmp[chromapix] on U and V) */
static int chroma_me_satd(x264o_encoder *e, int mbx, int mby, int ox, int oy, int w, int h, int ref, int mvx, int mvy)
{
    pixel pu[64], pv[64], fu[64], fv[64];
    const pixel *fuv = e->fenc_uv + (size_t)(mby * 8 + oy / 2) * e->fs + mbx * 16 + ox;
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    x264o_mc_chroma(pu, pv, 8, chroma_plane(e, ref), e->rs, mbx * 8 + ox / 2, mby * 8 + oy / 2, mvx, mvy, w / 2, h / 2);
    return x264o_satd(fu, 8, pu, 8, w / 2, h / 2) + x264o_satd(fv, 8, pv, 8, w / 2, h / 2);
}

/* Generic block search: w x h block at offset (ox,oy) inside macroblock (mbx,mby).  Start candidates are
 * tried in order (first-best wins), then hexagon + square full-pel search on SAD and the sub-pel diamonds
 * (half-pel on SAD, quarter-pel on SATD) — x264_me_search_ref + refine_subpel.  mvp = cost predictor. */
static me_result me_search_block(x264o_encoder *e, int mbx, int mby, int ox, int oy, int w, int h, int qp, int refidx,
                                 const int mvp[2], const int (*cand)[2], int ncand, int *halfpel_thresh)
{
    const pixel *fenc = e->fenc_y + (size_t)(mby * 16 + oy) * e->fs + mbx * 16 + ox;
    int ref = ref_slot(e, refidx);
    pixel *planes[4] = { luma_plane(e, ref, 0), luma_plane(e, ref, 1), luma_plane(e, ref, 2), luma_plane(e, ref, 3) };
    const pixel *full = planes[0] + (size_t)(mby * 16 + oy) * e->rs + mbx * 16 + ox;
    const uint16_t *cm = cost_mv_for(e, qp);
    int smin[2], smax[2], fmin[2], fmax[2];
    mv_limits(e, mbx, mby, smin, smax, fmin, fmax);
    const uint16_t *cmx = cm - mvp[0], *cmy = cm - mvp[1];
#define FPEL_COST(mx, my) (x264o_sad(fenc, e->fs, full + (my) * e->rs + (mx), e->rs, w, h) + cmx[(mx) * 4] + cmy[(my) * 4])
    int bmx = 0, bmy = 0, bcost = 1 << 28;
    for (int i = 0; i < ncand; i++) {
        int cx = clampi(cand[i][0], fmin[0], fmax[0]), cy = clampi(cand[i][1], fmin[1], fmax[1]);
        int c = FPEL_COST(cx, cy);
        if (c < bcost) { bcost = c; bmx = cx; bmy = cy; }
    }
    if (e->cfg.me_method == 3) {
        /* X264_ME_ESA ([x264-upstream] encoder/me.c): exhaustive search of the rectangle [bm - merange, bm + merange] clipped to the
         * full-pel limits, its width rounded up to a multiple of 4 as x264's successive-elimination rows are; raster order, a strictly
         * better candidate wins (the ADS / row-cost eliminations of x264 never drop a candidate that could win).  No hexagon / square
         * refine afterwards. */
        const int r = e->cfg.me_range;
        const int min_x = bmx - r > fmin[0] ? bmx - r : fmin[0], min_y = bmy - r > fmin[1] ? bmy - r : fmin[1];
        const int max_x = bmx + r < fmax[0] ? bmx + r : fmax[0], max_y = bmy + r < fmax[1] ? bmy + r : fmax[1];
        const int width = (max_x - min_x + 3) & ~3;
        for (int my = min_y; my <= max_y; my++)
            for (int mx = min_x; mx < min_x + width; mx++) {
                int c = FPEL_COST(mx, my);
                if (c < bcost) { bcost = c; bmx = mx; bmy = my; }
            }
    } else
    if (e->cfg.me_method == 0) {
        /* X264_ME_DIA ([x264-upstream] encoder/me.c): radius-1 diamond, up to merange steps; order (0,-1) (0,1) (-1,0) (1,0),
         * strictly-better wins, the centre wins ties; no square refine afterwards */
        static const int8_t dia1[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        int i = e->cfg.me_range;
        do {
            int best = -1;
            for (int k = 0; k < 4; k++) {
                int c = FPEL_COST(bmx + dia1[k][0], bmy + dia1[k][1]);
                if (c < bcost) { bcost = c; best = k; }
            }
            if (best < 0) break;
            bmx += dia1[best][0]; bmy += dia1[best][1];
        } while (--i && bmx >= fmin[0] && bmx <= fmax[0] && bmy >= fmin[1] && bmy <= fmax[1]);
    } else {
    int range = e->cfg.me_range;
    if (e->cfg.me_method == 2) {
        /* X264_ME_UMH ([x264-upstream] encoder/me.c, "Uneven-cross Multi-Hexagon-grid Search"): predictor diamonds, early
         * termination by SAD thresholds, uneven cross, 5x5 corners, 16-point hexagon grid rings with a search range adapted to
         * the SAD level and to the disagreement of the predictors, then the hexagon + square refine below with that range.
         * Every candidate update is x264's in-order "strictly better wins" (COPY3_IF_LT).
         * mvc (x264: the neighbours' vectors, without mvp): here the start candidates after the predictor for 16x16 (zero,
         * co-located), the 16x16 vector for sub-partitions. */
        static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
        static const int8_t hex4[16][2] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                            { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };
        const int shift = (w == 16 ? 0 : 1) + (h == 16 ? 0 : 1);            /* pixel_size_shift: 16x16 0, 16x8 / 8x16 1, 8x8 2 */
        const int pmx = clampi((mvp[0] + 2) >> 2, fmin[0], fmax[0]), pmy = clampi((mvp[1] + 2) >> 2, fmin[1], fmax[1]);
        int omx, omy, c_, done = 0, cross_start = 1;
#define INRANGE(mx, my) ((mx) >= fmin[0] && (mx) <= fmax[0] && (my) >= fmin[1] && (my) <= fmax[1])
#define COST_MV(mx, my) do { c_ = FPEL_COST(mx, my); if (c_ < bcost) { bcost = c_; bmx = (mx); bmy = (my); } } while (0)
#define COST_MV_X4(x0, y0, x1, y1, x2, y2, x3, y3) do { COST_MV(omx + (x0), omy + (y0)); COST_MV(omx + (x1), omy + (y1)); \
                                                        COST_MV(omx + (x2), omy + (y2)); COST_MV(omx + (x3), omy + (y3)); } while (0)
#define DIA1_ITER(mx, my) do { omx = (mx); omy = (my); COST_MV_X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
#define CROSS(start, x_max, y_max) do { \
            for (int i_ = (start); i_ < (x_max); i_ += 2) { \
                if (omx + i_ <= fmax[0]) COST_MV(omx + i_, omy); \
                if (omx - i_ >= fmin[0]) COST_MV(omx - i_, omy); } \
            for (int i_ = (start); i_ < (y_max); i_ += 2) { \
                if (omy + i_ <= fmax[1]) COST_MV(omx, omy + i_); \
                if (omy - i_ >= fmin[1]) COST_MV(omx, omy - i_); } } while (0)
#define SAD_THRESH(v) (bcost < ((v) >> shift))
        const int ucost1 = bcost;
        DIA1_ITER(pmx, pmy);
        if (pmx | pmy) DIA1_ITER(0, 0);
        const int ucost2 = bcost;
        if ((bmx | bmy) && ((bmx - pmx) | (bmy - pmy))) DIA1_ITER(bmx, bmy);
        if (bcost == ucost2) cross_start = 3;
        omx = bmx; omy = bmy;
        if (bcost == ucost2 && SAD_THRESH(2000)) {
            COST_MV_X4(0, -2, -1, -1, 1, -1, -2, 0);
            COST_MV_X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500)) done = 1;
            else if (bcost == ucost2) {
                const int r1 = (range >> 1) | 1;
                CROSS(3, r1, r1);
                COST_MV_X4(-1, -2, 1, -2, -2, -1, 2, -1);
                COST_MV_X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2) done = 1;
                cross_start = r1 + 2;
            }
        }
        if (!done) {
            /* adaptive search range: agreement of the predictors x SAD level */
            int mvd;
            if (w == 16 && h == 16) mvd = ncand <= 2 ? 25 : 4 * (abs(cand[1][0] - cand[2][0]) + abs(cand[1][1] - cand[2][1]));
            else mvd = abs(mvp[0] - 4 * cand[0][0]) + abs(mvp[1] - 4 * cand[0][1]);
            const int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            const int mvd_ctx = mvd < 10 ? 0 : mvd < 20 ? 1 : mvd < 40 ? 2 : 3;
            range = range * range_mul[mvd_ctx][sad_ctx] >> 2;
            /* x264 keeps the cross centred where the small diamonds left it ("FIXME ... is this desirable?") */
            CROSS(cross_start, range, range >> 1);
            COST_MV_X4(-2, -2, -2, 2, 2, -2, 2, 2);
            omx = bmx; omy = bmy;
            int i = 1;
            do {
                for (int j = 0; j < 16; j++) {
                    int mx = omx + hex4[j][0] * i, my = omy + hex4[j][1] * i;
                    if (INRANGE(mx, my)) COST_MV(mx, my);
                }
            } while (++i <= range >> 2);
            if (!INRANGE(bmx, bmy)) done = 1;
        }
#undef SAD_THRESH
#undef CROSS
#undef DIA1_ITER
#undef COST_MV_X4
#undef COST_MV
#undef INRANGE
        if (done) goto fullpel_done;
    }
    /* hexagon search (radius 2), then 3x3 square refine; first-best wins ties, centre wins over all */
    {
        int key = bcost << 3;
        for (int k = 1; k <= 6; k++) {
            int c = (FPEL_COST(bmx + hex2[k][0], bmy + hex2[k][1]) << 3) + k + 1;
            if (c < key) key = c;
        }
        if (key & 7) {
            int dir = (key & 7) - 2;
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (int i = (range >> 1) - 1; i > 0 && bmx >= fmin[0] && bmx <= fmax[0] && bmy >= fmin[1] && bmy <= fmax[1]; i--) {
                key &= ~7;
                for (int k = 0; k < 3; k++) {
                    int c = (FPEL_COST(bmx + hex2[dir + k][0], bmy + hex2[dir + k][1]) << 3) + k + 1;
                    if (c < key) key = c;
                }
                if (!(key & 7)) break;
                dir += (key & 7) - 2;
                dir = mod6m1[dir + 1];
                bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            }
        }
        bcost = key >> 3;
        int bdir = 0;
        for (int k = 1; k <= 8; k++) {
            int c = FPEL_COST(bmx + square1[k][0], bmy + square1[k][1]);
            if (c < bcost) { bcost = c; bdir = k; }
        }
        bmx += square1[bdir][0]; bmy += square1[bdir][1];
    }
    }
fullpel_done:;
#undef FPEL_COST
    /* sub-pel refinement (subme>=2): half-pel diamond on SAD, then quarter-pel diamond on SATD */
    static const uint8_t iters[12][2] = { { 0, 0 }, { 0, 0 }, { 1, 0 }, { 1, 0 }, { 1, 1 }, { 1, 2 }, { 2, 2 }, { 2, 2 }, { 4, 10 }, { 4, 10 }, { 4, 10 }, { 4, 10 } };
    int sub = clampi(e->cfg.subme, 0, 11);
    int mx = bmx * 4, my = bmy * 4;
    pixel pred[256];
    if (sub >= 2) {
        static const int8_t dia[4][2] = { { 0, -2 }, { 0, 2 }, { -2, 0 }, { 2, 0 } };
        for (int it = iters[sub][0]; it > 0; it--) {
            int omx = mx, omy = my;
            for (int k = 0; k < 4; k++) {
                int cx = omx + dia[k][0], cy = omy + dia[k][1];
                x264o_mc_luma(pred, 16, planes, e->rs, mbx * 16 + ox, mby * 16 + oy, cx, cy, w, h);
                int c = x264o_sad(fenc, e->fs, pred, 16, w, h) + cmx[cx] + cmy[cy];
                if (c < bcost) { bcost = c; mx = cx; my = cy; }
            }
            if (mx == omx && my == omy) break;
        }
        x264o_mc_luma(pred, 16, planes, e->rs, mbx * 16 + ox, mby * 16 + oy, mx, my, w, h);
        bcost = x264o_satd(fenc, e->fs, pred, 16, w, h) + cmx[mx] + cmy[my];
        /* b_chroma_me (subme >= 5, blocks of 8x8 and larger): every SATD cost that could become the best one also carries the
         * chroma SATD; a candidate whose luma cost alone is not below bcost cannot win, so x264 skips its chroma */
        const int chroma_me = e->cfg.chroma_me && sub >= 5;
        if (chroma_me) bcost += chroma_me_satd(e, mbx, mby, ox, oy, w, h, ref, mx, my);
        /* early termination when examining several reference frames ([x264-upstream] me.c refine_subpel, p_halfpel_thresh):
         * a reference whose half-pel SATD cost is more than 8/7 of the best one so far skips the quarter-pel diamond */
        int skip_qpel = 0;
        if (halfpel_thresh) {
            if (((bcost * 7) >> 3) > *halfpel_thresh) skip_qpel = 1;
            else if (bcost < *halfpel_thresh) *halfpel_thresh = bcost;
        }
        int bdir = -1;
        static const int8_t qd[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        for (int it = skip_qpel ? 0 : iters[sub][1]; it > 0; it--) {
            if (my <= smin[1] || my >= smax[1] || mx <= smin[0] || mx >= smax[0]) break;
            int odir = bdir, omx = mx, omy = my;
            for (int k = 0; k < 4; k++) {
                if ((k ^ 1) == odir) continue;     /* do not step back to where we came from */
                int cx = omx + qd[k][0], cy = omy + qd[k][1];
                x264o_mc_luma(pred, 16, planes, e->rs, mbx * 16 + ox, mby * 16 + oy, cx, cy, w, h);
                int c = x264o_satd(fenc, e->fs, pred, 16, w, h) + cmx[cx] + cmy[cy];
                if (chroma_me && c < bcost) c += chroma_me_satd(e, mbx, mby, ox, oy, w, h, ref, cx, cy);
                if (c < bcost) { bcost = c; mx = cx; my = cy; bdir = k; }
            }
            if (mx == omx && my == omy) break;
        }
    }
    me_result r = { mx, my, bcost };
    return r;
}

/* bits of ref_idx te(v) for `nref` active references */
static int ref_bits(int nref, int r) { return nref <= 1 ? 0 : nref == 2 ? 1 : bs_size_ue(r); }

static me_result me_search_16x16(x264o_encoder *e, int mbx, int mby, int qp, int refidx, int mvp[2], int *halfpel_thresh)
{
    /* start candidates, in priority order: predictor, zero, co-located previous-frame MV */
    int cand[3][2], ncand = 0, mi = mby * e->mbw + mbx;
    prev_mvp(e, mbx, mby, mvp);
    cand[ncand][0] = (mvp[0] + 2) >> 2; cand[ncand][1] = (mvp[1] + 2) >> 2; ncand++;
    cand[ncand][0] = 0; cand[ncand][1] = 0; ncand++;
    if (e->reff[0][mi] >= 0) { cand[ncand][0] = (e->mvf[0][mi][0] + 2) >> 2; cand[ncand][1] = (e->mvf[0][mi][1] + 2) >> 2; ncand++; }
    return me_search_block(e, mbx, mby, 0, 0, 16, 16, qp, refidx, mvp, (const int (*)[2])cand, ncand, halfpel_thresh);
}

/* intra 16x16 SATD estimate on SOURCE neighbours (lookahead-style; decides intra vs inter in P) */
static int intra16_estimate(x264o_encoder *e, int mbx, int mby, int lambda)
{
    const pixel *fenc = e->fenc_y + (size_t)mby * 16 * e->fs + mbx * 16;
    pixel pred[256];
    int left = mbx > 0, top = mby > 0, best = 1 << 28;
    int modes[4], n = 0;
    if (left && top) { modes[n++] = I_PRED_16x16_V; modes[n++] = I_PRED_16x16_H; modes[n++] = I_PRED_16x16_DC; modes[n++] = I_PRED_16x16_P; }
    else if (left) { modes[n++] = I_PRED_16x16_H; modes[n++] = I_PRED_16x16_DC_LEFT; }
    else if (top) { modes[n++] = I_PRED_16x16_V; modes[n++] = I_PRED_16x16_DC_TOP; }
    else modes[n++] = I_PRED_16x16_DC_128;
    for (int i = 0; i < n; i++) {
        int m = modes[i], sig = m > I_PRED_16x16_P ? I_PRED_16x16_DC : m;
        x264o_predict_16x16(pred, 16, fenc, e->fs, m);
        int c = x264o_satd(fenc, e->fs, pred, 16, 16, 16) + lambda * bs_size_ue(sig);
        if (c < best) best = c;
    }
    return best;
}

/* chroma counterpart of intra16_estimate for b_chroma_me: x264 adds i_satd_chroma to the intra costs it compares with inter costs
 * that carry chroma ([x264-upstream] analyse.c x264_macroblock_analyse, P slices).  Source neighbours, modes DC,H,V,P by availability. */
static int intra_chroma_estimate(x264o_encoder *e, int mbx, int mby, int lambda)
{
    const pixel *fuv = e->fenc_uv + (size_t)mby * 8 * e->fs + mbx * 16;
    pixel fu[64], fv[64], nu[9 * 9], nvv[9 * 9], pu[64], pv[64];
    int left = mbx > 0, top = mby > 0, best = 1 << 28;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    memset(nu, 128, sizeof(nu)); memset(nvv, 128, sizeof(nvv));
    for (int y = -1; y < 8; y++)
        for (int x = -1; x < 8; x++) {
            if (y >= 0 && x >= 0) continue;
            if ((y < 0 && !top) || (x < 0 && !left)) continue;
            nu[(y + 1) * 9 + x + 1] = fuv[y * e->fs + 2 * x]; nvv[(y + 1) * 9 + x + 1] = fuv[y * e->fs + 2 * x + 1];
        }
    int cm[4], cn = 0;
    if (left && top) { cm[cn++] = I_PRED_CHROMA_DC; cm[cn++] = I_PRED_CHROMA_H; cm[cn++] = I_PRED_CHROMA_V; cm[cn++] = I_PRED_CHROMA_P; }
    else if (left) { cm[cn++] = I_PRED_CHROMA_DC_LEFT; cm[cn++] = I_PRED_CHROMA_H; }
    else if (top) { cm[cn++] = I_PRED_CHROMA_DC_TOP; cm[cn++] = I_PRED_CHROMA_V; }
    else cm[cn++] = I_PRED_CHROMA_DC_128;
    for (int i = 0; i < cn; i++) {
        int m = cm[i], sig = m > I_PRED_CHROMA_P ? I_PRED_CHROMA_DC : m;
        x264o_predict_8x8c(pu, 8, nu + 10, 9, m);
        x264o_predict_8x8c(pv, 8, nvv + 10, 9, m);
        int c = x264o_satd(fu, 8, pu, 8, 8, 8) + x264o_satd(fv, 8, pv, 8, 8, 8) + lambda * bs_size_ue(sig);
        if (c < best) best = c;
    }
    return best;
}

/* Partition shapes of a P macroblock (D_16x16, D_16x8, D_8x16, D_8x8): block list per shape as
 * {ox, oy, w, h, first 8x8 index, second 8x8 index or -1} */
static const int8_t part_geom[4][4][6] = {
    { { 0, 0, 16, 16, 0, -1 } },
    { { 0, 0, 16, 8, 0, 1 }, { 0, 8, 16, 8, 2, 3 } },
    { { 0, 0, 8, 16, 0, 2 }, { 8, 0, 8, 16, 1, 3 } },
    { { 0, 0, 8, 8, 0, -1 }, { 8, 0, 8, 8, 1, -1 }, { 0, 8, 8, 8, 2, -1 }, { 8, 8, 8, 8, 3, -1 } } };
static const int8_t part_count[4] = { 1, 2, 2, 4 };
/* macroblock-type overhead in bits beyond P_L0_16x16: ue(1)/ue(2) = 3 bits, P_8x8 = ue(3) + 4 x ue(0) */
static const int8_t part_extra_bits[4] = { 0, 2, 2, 8 };

static void analyse_p_mb(x264o_encoder *e, int mbx, int mby, x264gpu_mb *mb)
{
    int mi = mby * e->mbw + mbx, qp = e->mbqp[mi], lambda = x264o_lambda(qp), mvp[2];
    /* 16x16 search in every usable reference (most recent first); lower index wins ties */
    me_result m = { 0, 0, 1 << 28 }, m16[4];
    int bref = 0;
    int halfpel_thresh = 1 << 28;          /* INT_MAX-like: shared by the references of this macroblock, only with more than one */
    for (int r = 0; r < e->nref; r++) {
        me_result t = me_search_16x16(e, mbx, mby, qp, r, mvp, e->nref > 1 ? &halfpel_thresh : NULL);
        m16[r] = t;
        t.cost += lambda * ref_bits(e->nref, r);
        if (t.cost < m.cost) { m = t; bref = r; }
    }
    int best_cost = m.cost, best_shape = 0;
    int best_mv[4][2] = { { m.mvx, m.mvy }, { m.mvx, m.mvy }, { m.mvx, m.mvy }, { m.mvx, m.mvy } };
    int best_ref[4] = { bref, bref, bref, bref };
    if ((e->cfg.partitions & 1) && e->cfg.mixed_refs && e->nref > 1) {
        /* --mixed-refs ([x264-upstream] analyse.c x264_mb_analyse_inter_p8x8_mixed_ref, _p16x8, _p8x16): every 8x8 block is
         * searched in every reference, starting from that reference's 16x16 vector, and keeps the cheapest (cost + ref bits,
         * lower index wins ties); 16x8 / 8x16 halves then try the references their two 8x8 blocks chose.  x264's early-out on
         * the neighbours' references needs raster-order neighbours and is not used. */
        int ref8[4], mv8[4][2], cost8[4], cost = lambda * part_extra_bits[3];
        /* x264's early termination ("if 16x16 chose ref 0, then evaluate no refs older than those used by the neighbors"), with
         * the PREVIOUS picture's field standing in for the raster-order neighbours as everywhere in this pipeline: left, top,
         * top-left, top-right and co-located macroblocks; needs a left and a top neighbour */
        int maxref = e->nref - 1;
        if (bref == 0 && mbx > 0 && mby > 0) {
            const int8_t *pr = e->reff[0];
            maxref = 0;
            const int nbs[5] = { mi - 1, mi - e->mbw, mi - e->mbw - 1, mbx + 1 < e->mbw ? mi - e->mbw + 1 : mi, mi };
            for (int i = 0; i < 5; i++) if (pr[nbs[i]] > maxref) maxref = pr[nbs[i]];
            if (maxref > e->nref - 1) maxref = e->nref - 1;
        }
        for (int p = 0; p < 4; p++) {
            const int8_t *g = part_geom[3][p];
            cost8[p] = 1 << 28;
            for (int r = 0; r <= maxref; r++) {
                int c0[1][2] = { { (m16[r].mvx + 2) >> 2, (m16[r].mvy + 2) >> 2 } };
                me_result t = me_search_block(e, mbx, mby, g[0], g[1], g[2], g[3], qp, r, mvp, (const int (*)[2])c0, 1, NULL);
                t.cost += lambda * ref_bits(e->nref, r);
                if (t.cost < cost8[p]) { cost8[p] = t.cost; ref8[p] = r; mv8[p][0] = t.mvx; mv8[p][1] = t.mvy; }
            }
            cost += cost8[p];
        }
        if (cost < best_cost) { best_cost = cost; best_shape = 3; memcpy(best_mv, mv8, sizeof(mv8)); memcpy(best_ref, ref8, sizeof(ref8)); }
        for (int shape = 1; shape <= 2 && best_shape != 0; shape++) {
            int mv[4][2], rf[4];
            cost = lambda * part_extra_bits[shape];
            for (int p = 0; p < 2; p++) {
                const int8_t *g = part_geom[shape][p];
                const int cand_ref[2] = { ref8[g[4]], ref8[g[5]] };
                int bc = 1 << 28;
                for (int i = 0; i < (cand_ref[0] == cand_ref[1] ? 1 : 2); i++) {
                    const int r = cand_ref[i];
                    int c0[1][2] = { { (m16[r].mvx + 2) >> 2, (m16[r].mvy + 2) >> 2 } };
                    me_result t = me_search_block(e, mbx, mby, g[0], g[1], g[2], g[3], qp, r, mvp, (const int (*)[2])c0, 1, NULL);
                    t.cost += lambda * ref_bits(e->nref, r);
                    if (t.cost < bc) { bc = t.cost; mv[g[4]][0] = mv[g[5]][0] = t.mvx; mv[g[4]][1] = mv[g[5]][1] = t.mvy; rf[g[4]] = rf[g[5]] = r; }
                }
                cost += bc;
            }
            if (cost < best_cost) { best_cost = cost; best_shape = shape; memcpy(best_mv, mv, sizeof(mv)); memcpy(best_ref, rf, sizeof(rf)); }
        }
    } else
    if (e->cfg.partitions & 1) {
        /* sub-partition searches start from the 16x16 vector, in the 16x16 winner's reference (no mixed refs);
         * 16x8 / 8x16 only when 8x8 beats 16x16 */
        int c0[1][2] = { { (m.mvx + 2) >> 2, (m.mvy + 2) >> 2 } };
        static const int order[3] = { 3, 1, 2 };
        for (int oi = 0; oi < 3; oi++) {
            int shape = order[oi], cost = lambda * (part_extra_bits[shape] + part_count[shape] * ref_bits(e->nref, bref)), mv[4][2];
            if (oi > 0 && best_shape == 0) break;
            for (int p = 0; p < part_count[shape]; p++) {
                const int8_t *g = part_geom[shape][p];
                me_result r = me_search_block(e, mbx, mby, g[0], g[1], g[2], g[3], qp, bref, mvp, (const int (*)[2])c0, 1, NULL);
                cost += r.cost;
                mv[g[4]][0] = r.mvx; mv[g[4]][1] = r.mvy;
                if (g[5] >= 0) { mv[g[5]][0] = r.mvx; mv[g[5]][1] = r.mvy; }
            }
            if (cost < best_cost) { best_cost = cost; best_shape = shape; memcpy(best_mv, mv, sizeof(mv)); }
        }
    }
    int icost = intra16_estimate(e, mbx, mby, lambda);
    if (e->cfg.chroma_me && e->cfg.subme >= 5) icost += intra_chroma_estimate(e, mbx, mby, lambda);
    memset(mb, 0, sizeof(*mb));
    mb->qp = (uint8_t)qp;
    mb->aux[0] = best_cost; mb->aux[1] = icost; mb->aux[2] = m.cost;
    /* the field used as next frame's predictor always carries the 16x16 vector */
    if (icost < best_cost) {
        mb->type = X264GPU_MB_I16x16;   /* provisional: real intra analysis happens in the intra stage */
        mb->cost = icost;
        e->reff[1][mi] = -1; e->mvf[1][mi][0] = e->mvf[1][mi][1] = 0;
        for (int k = 0; k < 4; k++) mb->ref[k] = -1;
    } else {
        mb->type = best_shape == 3 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0;
        mb->partition = (uint8_t)best_shape;
        mb->cost = best_cost;
        for (int k = 0; k < 4; k++) { mb->mv[k][0] = (int16_t)best_mv[k][0]; mb->mv[k][1] = (int16_t)best_mv[k][1]; mb->ref[k] = (int8_t)best_ref[k]; }
        /* the ref field keeps the oldest reference the macroblock uses (>= 0: inter); the vector field the 16x16 vector */
        int oldest = best_ref[0];
        for (int k = 1; k < 4; k++) if (best_ref[k] > oldest) oldest = best_ref[k];
        e->reff[1][mi] = (int8_t)oldest; e->mvf[1][mi][0] = (int16_t)m.mvx; e->mvf[1][mi][1] = (int16_t)m.mvy;
    }
}

/* ---- stage 2: inter macroblock encode (x264_macroblock_encode, P_L0 16x16) ---- */
static void encode_inter_mb(x264o_encoder *e, int mbx, int mby, x264gpu_mb *mb, int16_t *lv)
{
    int qp = mb->qp, qpc = x264o_chroma_qp[clampi(qp + e->cfg.chroma_qp_offset, 0, 51)];
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)mby * 16 * e->rs + mbx * 16;
    pixel *rec_uv = chroma_plane(e, e->cur) + (size_t)mby * 8 * e->rs + mbx * 16;
    pixel pu[64], pv[64];
    for (int k = 0; k < 4; k++) {      /* motion compensation per 8x8 quadrant (covers 16x16 / 16x8 / 8x16 / 8x8) */
        int ox = (k & 1) * 8, oy = (k >> 1) * 8, ref = ref_slot(e, mb->ref[k]);       /* the reference is per 8x8 block (mixed refs) */
        pixel *planes[4] = { luma_plane(e, ref, 0), luma_plane(e, ref, 1), luma_plane(e, ref, 2), luma_plane(e, ref, 3) };
        x264o_mc_luma(rec + oy * e->rs + ox, e->rs, planes, e->rs, mbx * 16 + ox, mby * 16 + oy, mb->mv[k][0], mb->mv[k][1], 8, 8);
        x264o_mc_chroma(pu + (oy / 2) * 8 + ox / 2, pv + (oy / 2) * 8 + ox / 2, 8, chroma_plane(e, ref), e->rs, mbx * 8 + ox / 2, mby * 8 + oy / 2,
                        mb->mv[k][0], mb->mv[k][1], 4, 4);
    }
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) { rec_uv[y * e->rs + 2 * x] = pu[y * 8 + x]; rec_uv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
    memset(lv, 0, X264GPU_MB_LEVELS * sizeof(int16_t));
    const pixel *fenc = e->fenc_y + (size_t)mby * 16 * e->fs + mbx * 16;
    /* transform size ([x264-upstream] analyse.c x264_mb_analyse_transform): SA8D vs SATD of the prediction error */
    mb->transform8x8 = 0;
    if (e->cfg.dct8x8) mb->transform8x8 = x264o_sa8d(fenc, e->fs, rec, e->rs, 16, 16) < x264o_satd(fenc, e->fs, rec, e->rs, 16, 16);
    if (mb->transform8x8) encode_luma_inter8(e, fenc, rec, qp, mb, lv);
    else encode_luma_inter(e, fenc, rec, qp, mb, lv);
    if (!mb->cbp_luma) mb->transform8x8 = 0;      /* the flag is not transmitted without luma coefficients (macroblock_cache_save) */
    encode_chroma(e, e->fenc_uv + (size_t)mby * 8 * e->fs + mbx * 16, rec_uv, qpc, 1, mb, lv);
}

/* ---- stage 3: intra macroblock analysis + encode with reconstructed neighbours ---- */
static int i4_pred_mode(const x264gpu_mb *mbs, int mbw, int mbx, int mby, int b, const uint8_t *cur_modes)
{
    /* 8.3.1.1 / 8.3.2.1: min of left/top block modes; DC when a neighbour is absent; neighbour MBs that are not I_NxN
     * count as DC.  I8x8 macroblocks store each 8x8 mode replicated over its four 4x4 entries, which makes the 4x4
     * look-up of the top-left 4x4 of an 8x8 block exactly predIntra8x8PredMode (x264_mb_predict_intra4x4_mode(h, 4*idx)). */
    int bx = blk_x[b], by = blk_y[b], ma, mb_;
    static const uint8_t idx_of[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };
    if (bx > 0) ma = cur_modes[idx_of[by][bx - 1]];
    else if (mbx > 0) { const x264gpu_mb *n = &mbs[mby * mbw + mbx - 1]; ma = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[idx_of[by][3]] : 2; }
    else return 2;
    if (by > 0) mb_ = cur_modes[idx_of[by - 1][bx]];
    else if (mby > 0) { const x264gpu_mb *n = &mbs[(mby - 1) * mbw + mbx]; mb_ = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[idx_of[3][bx]] : 2; }
    else return 2;
    return ma < mb_ ? ma : mb_;
}

static int i4_avail(int mbx, int mby, int mbw, int b)
{
    int bx = blk_x[b], by = blk_y[b], a = 0;
    static const uint8_t idx_of[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };
    if (bx > 0 || mbx > 0) a |= X264O_AVAIL_LEFT;
    if (by > 0 || mby > 0) a |= X264O_AVAIL_TOP;
    if ((bx > 0 || mbx > 0) && (by > 0 || mby > 0)) a |= X264O_AVAIL_TOPLEFT;
    if (by == 0) { if (mby > 0 && (bx < 3 || mbx + 1 < mbw)) a |= X264O_AVAIL_TOPRIGHT; }
    else if (bx < 3 && idx_of[by - 1][bx + 1] < b) a |= X264O_AVAIL_TOPRIGHT;
    return a;
}

static void intra_mb(x264o_encoder *e, int mbx, int mby, int qp, x264gpu_mb *mbs, int16_t *lv)
{
    x264gpu_mb *mb = &mbs[mby * e->mbw + mbx];
    int lambda = x264o_lambda(qp), qpc = x264o_chroma_qp[clampi(qp + e->cfg.chroma_qp_offset, 0, 51)];
    const pixel *fenc = e->fenc_y + (size_t)mby * 16 * e->fs + mbx * 16;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)mby * 16 * e->rs + mbx * 16;
    int left = mbx > 0, top = mby > 0;
    pixel pred[256];
    int aux0 = mb->aux[0], aux1 = mb->aux[1], aux2 = mb->aux[2];     /* keep the P-slice analysis diagnostics */
    memset(mb, 0, sizeof(*mb));
    memset(lv, 0, X264GPU_MB_LEVELS * sizeof(int16_t));
    mb->qp = (uint8_t)qp;
    if (e->slice_type == X264GPU_SLICE_P) { mb->aux[0] = aux0; mb->aux[1] = aux1; mb->aux[2] = aux2; }
    for (int k = 0; k < 4; k++) mb->ref[k] = -1;
    /* --- intra 16x16 mode decision (SATD + lambda*ue(mode)); order V,H,DC,P, first-best wins --- */
    int modes[4], n = 0, best16 = 1 << 28, mode16 = 0;
    if (left && top) { modes[n++] = I_PRED_16x16_V; modes[n++] = I_PRED_16x16_H; modes[n++] = I_PRED_16x16_DC; modes[n++] = I_PRED_16x16_P; }
    else if (left) { modes[n++] = I_PRED_16x16_H; modes[n++] = I_PRED_16x16_DC_LEFT; }
    else if (top) { modes[n++] = I_PRED_16x16_V; modes[n++] = I_PRED_16x16_DC_TOP; }
    else modes[n++] = I_PRED_16x16_DC_128;
    for (int i = 0; i < n; i++) {
        int m = modes[i], sig = m > I_PRED_16x16_P ? I_PRED_16x16_DC : m;
        x264o_predict_16x16(pred, 16, rec, e->rs, m);
        int c = x264o_satd(fenc, e->fs, pred, 16, 16, 16) + lambda * bs_size_ue(sig);
        if (c < best16) { best16 = c; mode16 = m; }
    }
    /* --- intra 4x4: per block 9 modes on reconstructed neighbours, coded as we go --- */
    int use_i4 = 0;
    const int parts = (e->slice_type == X264GPU_SLICE_I && (e->cfg.partitions & 0x100)) ? (e->cfg.partitions >> 8) & 6 : e->cfg.partitions & 7;
    if (parts & 2) {
        pixel save[256];
        for (int y = 0; y < 16; y++) memcpy(save + y * 16, rec + y * e->rs, 16);
        int cost4 = lambda * (24 + 16);
        uint8_t m4[16];
        int16_t lv4[256];
        uint32_t nnz4 = 0;
        const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4IY][qp], *bias = e->qt.quant4_bias[X264O_CQM_4IY][qp];
        for (int b = 0; b < 16; b++) {
            const pixel *f = fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4;
            pixel *r = rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4;
            int avail = i4_avail(mbx, mby, e->mbw, b), pm = i4_pred_mode(mbs, e->mbw, mbx, mby, b, m4);
            int bestc = 1 << 28, bestm = 2;
            pixel p4[16], bp[16];
            for (int m = 0; m < 9; m++) {
                int real = m;
                int need_l = m == I_PRED_4x4_H || m == I_PRED_4x4_HU, need_t = m == I_PRED_4x4_V || m == I_PRED_4x4_DDL || m == I_PRED_4x4_VL;
                int need_all = m == I_PRED_4x4_DDR || m == I_PRED_4x4_VR || m == I_PRED_4x4_HD;
                if (need_l && !(avail & X264O_AVAIL_LEFT)) continue;
                if (need_t && !(avail & X264O_AVAIL_TOP)) continue;
                if (need_all && (avail & (X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPLEFT)) != (X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPLEFT)) continue;
                if (m == I_PRED_4x4_DC) {
                    int l = avail & X264O_AVAIL_LEFT, t = avail & X264O_AVAIL_TOP;
                    real = l && t ? I_PRED_4x4_DC : l ? I_PRED_4x4_DC_LEFT : t ? I_PRED_4x4_DC_TOP : I_PRED_4x4_DC_128;
                }
                x264o_predict_4x4(p4, 4, r, e->rs, real, avail);
                int c = x264o_satd(f, e->fs, p4, 4, 4, 4) + (m == pm ? 0 : 3 * lambda);
                if (c < bestc) { bestc = c; bestm = m; memcpy(bp, p4, 16); }
            }
            cost4 += bestc;
            m4[b] = (uint8_t)bestm;
            for (int y = 0; y < 4; y++) memcpy(r + y * e->rs, bp + y * 4, 4);
            dctcoef d[16];
            x264o_sub4x4_dct(d, f, e->fs, r, e->rs);
            if (x264o_quant_4x4(d, mf, bias)) {
                scan4(lv4 + b * 16, d);
                x264o_dequant_4x4(d, e->qt.dequant4_mf, qp);
                x264o_add4x4_idct(r, e->rs, d);
                nnz4 |= 1u << b;
            } else memset(lv4 + b * 16, 0, 32);
            if (cost4 >= best16) break;          /* early termination: i4x4 can no longer beat i16x16 */
        }
        if (cost4 < best16) {
            use_i4 = 1;
            mb->type = X264GPU_MB_I4x4;
            mb->cost = cost4;
            memcpy(mb->i4_mode, m4, 16);
            memcpy(lv, lv4, sizeof(lv4));
            mb->nnz = nnz4;
            for (int b = 0; b < 16; b++) if (nnz4 >> b & 1) mb->cbp_luma |= 1 << (b >> 2);
        } else
            for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, save + y * 16, 16);
    }
    /* --- intra 8x8 ([x264-upstream] analyse.c x264_mb_analyse_intra, I8x8 branch + x264_mb_encode_i8x8): per 8x8 block
     *     9 modes on the 1-2-1 filtered edge, cost = SA8D + 3*lambda unless the mode is the predicted one, base 4*lambda;
     *     blocks are coded as we go.  x264 analyses 8x8 before 4x4; the outcome does not depend on the order (the early
     *     terminations only reject candidates that lose the final strict-< comparison anyway), so it runs last here and
     *     is selected only if strictly cheaper than the best of 16x16 / 4x4 — x264's COPY2_IF_LT order. --- */
    int use_i8 = 0;
    if ((parts & 4) && e->cfg.dct8x8) {
        pixel save[256];
        for (int y = 0; y < 16; y++) memcpy(save + y * 16, rec + y * e->rs, 16);
        const int cur = use_i4 ? mb->cost : best16;
        int cost8 = lambda * 4, done = 1;
        uint8_t m8[16];
        int16_t lv8[256];
        uint32_t nnz8 = 0;
        int cbp8 = 0;
        memset(lv8, 0, sizeof(lv8));
        memset(m8, 2, sizeof(m8));
        const uint16_t *mf = e->qt.quant8_mf[X264O_CQM_8IY][qp], *bias = e->qt.quant8_bias[X264O_CQM_8IY][qp];
        for (int i8 = 0; i8 < 4; i8++) {
            int x8 = i8 & 1, y8 = i8 >> 1, avail = 0;
            const pixel *f = fenc + y8 * 8 * e->fs + x8 * 8;
            pixel *r = rec + y8 * 8 * e->rs + x8 * 8;
            if (x8 || left) avail |= X264O_AVAIL_LEFT;
            if (y8 || top) avail |= X264O_AVAIL_TOP;
            if ((x8 || left) && (y8 || top)) avail |= X264O_AVAIL_TOPLEFT;
            if (i8 == 0 ? top : i8 == 1 ? (top && mbx + 1 < e->mbw) : i8 == 2) avail |= X264O_AVAIL_TOPRIGHT;
            int pm = i4_pred_mode(mbs, e->mbw, mbx, mby, i8 * 4, m8);
            pixel edge[33], p8[64], bp[64];
            x264o_predict_8x8_filter(r, e->rs, edge, avail);
            int bestc = 1 << 28, bestm = 2;
            for (int m = 0; m < 9; m++) {
                int real = m;
                int need_l = m == I_PRED_4x4_H || m == I_PRED_4x4_HU, need_t = m == I_PRED_4x4_V || m == I_PRED_4x4_DDL || m == I_PRED_4x4_VL;
                int need_all = m == I_PRED_4x4_DDR || m == I_PRED_4x4_VR || m == I_PRED_4x4_HD;
                if (need_l && !(avail & X264O_AVAIL_LEFT)) continue;
                if (need_t && !(avail & X264O_AVAIL_TOP)) continue;
                if (need_all && (avail & (X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPLEFT)) != (X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPLEFT)) continue;
                if (m == I_PRED_4x4_DC) {
                    int l = avail & X264O_AVAIL_LEFT, t = avail & X264O_AVAIL_TOP;
                    real = l && t ? I_PRED_4x4_DC : l ? I_PRED_4x4_DC_LEFT : t ? I_PRED_4x4_DC_TOP : I_PRED_4x4_DC_128;
                }
                x264o_predict_8x8(p8, 8, edge, real);
                int c = x264o_sa8d(f, e->fs, p8, 8, 8, 8) + (m == pm ? 0 : 3 * lambda);
                if (c < bestc) { bestc = c; bestm = m; memcpy(bp, p8, 64); }
            }
            cost8 += bestc;
            memset(m8 + i8 * 4, bestm, 4);
            if (i8 < 3 && cost8 > cur) { done = 0; break; }     /* cannot win any more */
            for (int y = 0; y < 8; y++) memcpy(r + y * e->rs, bp + y * 8, 8);
            dctcoef d[64];
            x264o_sub8x8_dct8(d, f, e->fs, r, e->rs);
            if (x264o_quant_8x8(d, mf, bias)) {
                for (int k = 0; k < 64; k++) {
                    int16_t v = d[x264o_zigzag8[k]];
                    lv8[(i8 * 4 + (k & 3)) * 16 + (k >> 2)] = v;
                    if (v) nnz8 |= 1u << (i8 * 4 + (k & 3));
                }
                x264o_dequant_8x8(d, e->qt.dequant8_mf, qp);
                x264o_add8x8_idct8(r, e->rs, d);
                cbp8 |= 1 << i8;
            }
        }
        if (done && cost8 < cur) {
            use_i8 = 1; use_i4 = 0;
            mb->type = X264GPU_MB_I8x8;
            mb->cost = cost8;
            mb->transform8x8 = 1;
            memcpy(mb->i4_mode, m8, 16);
            memcpy(lv, lv8, sizeof(lv8));
            mb->nnz = nnz8;
            mb->cbp_luma = (uint8_t)cbp8;
        } else
            for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, save + y * 16, 16);
    }
    if (!use_i4 && !use_i8) {
        /* x264_mb_encode_i16x16 */
        mb->type = X264GPU_MB_I16x16;
        mb->cost = best16;
        mb->i16_mode = (uint8_t)(mode16 > I_PRED_16x16_P ? I_PRED_16x16_DC : mode16);
        x264o_predict_16x16(pred, 16, rec, e->rs, mode16);
        for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, pred + y * 16, 16);
        const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4IY][qp], *bias = e->qt.quant4_bias[X264O_CQM_4IY][qp];
        dctcoef d[16][16], dc[16];
        int nz[16], any_ac = 0;
        for (int b = 0; b < 16; b++) {
            x264o_sub4x4_dct(d[b], fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4, e->fs, rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs);
            dc[blk_y[b] * 4 + blk_x[b]] = d[b][0]; d[b][0] = 0;
            nz[b] = x264o_quant_4x4(d[b], mf, bias);
            scan4(lv + b * 16, d[b]);
            if (nz[b]) { any_ac = 1; mb->nnz |= 1u << b; x264o_dequant_4x4(d[b], e->qt.dequant4_mf, qp); }
        }
        mb->cbp_luma = any_ac ? 15 : 0;
        x264o_dct4x4dc(dc);
        int nzdc = x264o_quant_4x4_dc(dc, mf[0] >> 1, bias[0] << 1);
        scan4(lv + X264GPU_LV_LUMA_DC, dc);
        if (nzdc) { mb->nnz |= 1u << 24; x264o_idct4x4dc(dc); x264o_dequant_4x4_dc(dc, e->qt.dequant4_mf, qp); }
        for (int b = 0; b < 16; b++) {
            d[b][0] = nzdc ? dc[blk_y[b] * 4 + blk_x[b]] : 0;
            x264o_add4x4_idct(rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs, d[b]);
        }
    }
    /* --- chroma: mode by SATD(U)+SATD(V)+lambda*ue(mode), order DC,H,V,P --- */
    const pixel *fuv = e->fenc_uv + (size_t)mby * 8 * e->fs + mbx * 16;
    pixel *ruv = chroma_plane(e, e->cur) + (size_t)mby * 8 * e->rs + mbx * 16;
    pixel fu[64], fv[64], nu[9 * 9], nvv[9 * 9], pu[64], pv[64], bu[64], bv[64];
    /* de-interleave source and the neighbour ring (row -1 / col -1) into planar scratch with stride 9 */
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    memset(nu, 128, sizeof(nu)); memset(nvv, 128, sizeof(nvv));
    for (int y = -1; y < 8; y++)
        for (int x = -1; x < 8; x++) {
            if (y >= 0 && x >= 0) continue;
            if ((y < 0 && !top) || (x < 0 && !left)) continue;
            nu[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x]; nvv[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x + 1];
        }
    int cm[4], cn = 0, bestc = 1 << 28, modec = 0;
    if (left && top) { cm[cn++] = I_PRED_CHROMA_DC; cm[cn++] = I_PRED_CHROMA_H; cm[cn++] = I_PRED_CHROMA_V; cm[cn++] = I_PRED_CHROMA_P; }
    else if (left) { cm[cn++] = I_PRED_CHROMA_DC_LEFT; cm[cn++] = I_PRED_CHROMA_H; }
    else if (top) { cm[cn++] = I_PRED_CHROMA_DC_TOP; cm[cn++] = I_PRED_CHROMA_V; }
    else cm[cn++] = I_PRED_CHROMA_DC_128;
    for (int i = 0; i < cn; i++) {
        int m = cm[i], sig = m > I_PRED_CHROMA_P ? I_PRED_CHROMA_DC : m;
        x264o_predict_8x8c(pu, 8, nu + 10, 9, m);
        x264o_predict_8x8c(pv, 8, nvv + 10, 9, m);
        int c = x264o_satd(fu, 8, pu, 8, 8, 8) + x264o_satd(fv, 8, pv, 8, 8, 8) + lambda * bs_size_ue(sig);
        if (c < bestc) { bestc = c; modec = sig; memcpy(bu, pu, 64); memcpy(bv, pv, 64); }
    }
    mb->chroma_mode = (uint8_t)modec;
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { ruv[y * e->rs + 2 * x] = bu[y * 8 + x]; ruv[y * e->rs + 2 * x + 1] = bv[y * 8 + x]; }
    encode_chroma(e, fuv, ruv, qpc, 0, mb, lv);
}

/* ---- stage 4: deblocking of the whole frame in macroblock raster order (8.7) ---- */
static int blk_nnz(const x264gpu_mb *m, int bx, int by)
{
    static const uint8_t idx_of[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };
    if (m->type == X264GPU_MB_I16x16) return 1;   /* intra: bS >= 3 anyway */
    if (m->transform8x8) return (m->cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1;   /* 8.7.2.1: the 8x8 block containing the sample */
    return (m->nnz >> idx_of[by][bx]) & 1;
}
static int is_intra(const x264gpu_mb *m) { return m->type == X264GPU_MB_I4x4 || m->type == X264GPU_MB_I8x8 || m->type == X264GPU_MB_I16x16; }

static int edge_bs(const x264gpu_mb *p, int pbx, int pby, const x264gpu_mb *q, int qbx, int qby, int mb_edge)
{
    if (is_intra(p) || is_intra(q)) return mb_edge ? 4 : 3;
    if (blk_nnz(p, pbx, pby) || blk_nnz(q, qbx, qby)) return 2;
    int pi = (pby >> 1) * 2 + (pbx >> 1), qi = (qby >> 1) * 2 + (qbx >> 1);
    if (p->ref[pi] != q->ref[qi]) return 1;
    if (abs(p->mv[pi][0] - q->mv[qi][0]) >= 4 || abs(p->mv[pi][1] - q->mv[qi][1]) >= 4) return 1;
    return 0;
}

static void deblock_frame(x264o_encoder *e, const x264gpu_mb *mbs)
{
    int a_off = e->cfg.deblock_alpha * 2, b_off = e->cfg.deblock_beta * 2;   /* slice_alpha_c0_offset_div2 * 2 */
    pixel *Y = luma_plane(e, e->cur, 0), *UV = chroma_plane(e, e->cur);
    for (int mby = 0; mby < e->mbh; mby++)
        for (int mbx = 0; mbx < e->mbw; mbx++) {
            const x264gpu_mb *q = &mbs[mby * e->mbw + mbx];
            for (int dir = 0; dir < 2; dir++)          /* 0: vertical edges (filter across x), 1: horizontal */
                for (int edge = 0; edge < 4; edge++) {
                    const x264gpu_mb *p = q;
                    if ((edge & 1) && q->transform8x8) continue;     /* no transform edge at 4-sample offsets */
                    if (edge == 0) {
                        if (dir == 0) { if (mbx == 0) continue; p = &mbs[mby * e->mbw + mbx - 1]; }
                        else { if (mby == 0) continue; p = &mbs[(mby - 1) * e->mbw + mbx]; }
                    }
                    int qpav = (p->qp + q->qp + 1) >> 1;
                    int qpc_p = x264o_chroma_qp[clampi(p->qp + e->cfg.chroma_qp_offset, 0, 51)];
                    int qpc_q = x264o_chroma_qp[clampi(q->qp + e->cfg.chroma_qp_offset, 0, 51)];
                    int qpcav = (qpc_p + qpc_q + 1) >> 1;
                    int ia = clampi(qpav + a_off, 0, 51), ib = clampi(qpav + b_off, 0, 51);
                    int ica = clampi(qpcav + a_off, 0, 51), icb = clampi(qpcav + b_off, 0, 51);
                    for (int k = 0; k < 4; k++) {      /* four 4-sample segments along the edge */
                        int qbx = dir == 0 ? edge : k, qby = dir == 0 ? k : edge;
                        int pbx = dir == 0 ? (edge + 3) & 3 : k, pby = dir == 0 ? k : (edge + 3) & 3;
                        int bs = edge_bs(p, pbx, pby, q, qbx, qby, edge == 0);
                        if (!bs) continue;
                        int tc0 = bs < 4 ? x264o_tc0_table[ia][bs - 1] : 0;
                        pixel *py = Y + (size_t)(mby * 16 + qby * 4) * e->rs + mbx * 16 + qbx * 4;
                        if (dir == 0) x264o_deblock_luma_edge(py, 1, e->rs, 4, x264o_alpha_table[ia], x264o_beta_table[ib], tc0, bs);
                        else x264o_deblock_luma_edge(py, e->rs, 1, 4, x264o_alpha_table[ia], x264o_beta_table[ib], tc0, bs);
                        if (!(edge & 1)) {             /* chroma edges live on luma edges 0 and 2 */
                            int ctc0 = bs < 4 ? x264o_tc0_table[ica][bs - 1] : 0;
                            for (int c = 0; c < 2; c++) {
                                pixel *pc = UV + (size_t)(mby * 8 + qby * 2) * e->rs + 2 * (mbx * 8 + qbx * 2) + c;
                                if (dir == 0) x264o_deblock_chroma_edge(pc, 2, e->rs, 2, x264o_alpha_table[ica], x264o_beta_table[icb], ctc0, bs);
                                else x264o_deblock_chroma_edge(pc, e->rs, 2, 2, x264o_alpha_table[ica], x264o_beta_table[icb], ctc0, bs);
                            }
                        }
                    }
                }
        }
}

/* ---- stage 5: half-pel planes + border expansion of the new reference (A4) ---- */
static void filter_frame(x264o_encoder *e)
{
    pixel *planes[4] = { luma_plane(e, e->cur, 0), luma_plane(e, e->cur, 1), luma_plane(e, e->cur, 2), luma_plane(e, e->cur, 3) };
    x264o_frame_filter(planes, e->rs, e->cw, e->ch, PAD);
    pixel *uv = chroma_plane(e, e->cur);
    int cw = e->cw / 2, chh = e->ch / 2;
    for (int y = -CPAD; y < chh + CPAD; y++)
        for (int x = -CPAD; x < cw + CPAD; x++)
            if (x < 0 || x >= cw || y < 0 || y >= chh) {
                int sx = clampi(x, 0, cw - 1), sy = clampi(y, 0, chh - 1);
                uv[(size_t)y * e->rs + 2 * x] = uv[(size_t)sy * e->rs + 2 * sx];
                uv[(size_t)y * e->rs + 2 * x + 1] = uv[(size_t)sy * e->rs + 2 * sx + 1];
            }
}

/* stages: bit i set = run stage i (0 ingest,1 analyse,2 inter,3 intra,4 deblock,5 filter); tests use
 * partial runs to localise GPU mismatches.  Normal use: stages = 0x3f. */
int x264o_encoder_encode(x264o_encoder *e, const uint8_t *i420, int slice_type, x264gpu_mb *mbs, int16_t *levels)
{
    int n = e->mbw * e->mbh;
    if (slice_type == X264GPU_SLICE_P && !e->have_ref) return -1;
    if (slice_type == X264GPU_SLICE_I) e->have_ref = 0;      /* IDR: the DPB is emptied */
    if (slice_type == X264GPU_SLICE_I_NONIDR) slice_type = X264GPU_SLICE_I;      /* same coding tools, the references stay */
    e->slice_type = slice_type;
    e->nref = e->have_ref < e->slots - 1 ? e->have_ref : e->slots - 1;
    ingest(e, i420);
    const int slice_qp = slice_type == X264GPU_SLICE_I ? e->cfg.qp_i : e->cfg.qp_p;
    compute_mb_qp(e, slice_qp);
    if (slice_type == X264GPU_SLICE_I) {
        for (int i = 0; i < n; i++) { e->reff[1][i] = -1; e->mvf[1][i][0] = e->mvf[1][i][1] = 0; }
        for (int mby = 0; mby < e->mbh; mby++)
            for (int mbx = 0; mbx < e->mbw; mbx++)
                intra_mb(e, mbx, mby, e->mbqp[mby * e->mbw + mbx], mbs, levels + (size_t)(mby * e->mbw + mbx) * X264GPU_MB_LEVELS);
    } else {
        for (int mby = 0; mby < e->mbh; mby++)
            for (int mbx = 0; mbx < e->mbw; mbx++) analyse_p_mb(e, mbx, mby, &mbs[mby * e->mbw + mbx]);
        for (int mby = 0; mby < e->mbh; mby++)
            for (int mbx = 0; mbx < e->mbw; mbx++) {
                x264gpu_mb *mb = &mbs[mby * e->mbw + mbx];
                if (mb->type == X264GPU_MB_P_L0 || mb->type == X264GPU_MB_P_8x8) encode_inter_mb(e, mbx, mby, mb, levels + (size_t)(mby * e->mbw + mbx) * X264GPU_MB_LEVELS);
            }
        for (int mby = 0; mby < e->mbh; mby++)
            for (int mbx = 0; mbx < e->mbw; mbx++) {
                x264gpu_mb *mb = &mbs[mby * e->mbw + mbx];
                if (mb->type != X264GPU_MB_P_L0 && mb->type != X264GPU_MB_P_8x8) intra_mb(e, mbx, mby, e->mbqp[mby * e->mbw + mbx], mbs, levels + (size_t)(mby * e->mbw + mbx) * X264GPU_MB_LEVELS);
            }
    }
    if (e->cfg.aq_mode || e->ext_off_q8) settle_mb_qp(e, mbs, slice_qp);
    if (e->cfg.deblock) deblock_frame(e, mbs);
    filter_frame(e);
    /* rotate: the frame just built becomes the reference; its MV field becomes "previous" */
    e->cur = (e->cur + 1) % e->slots;
    { int16_t (*t)[2] = e->mvf[0]; e->mvf[0] = e->mvf[1]; e->mvf[1] = t; }
    { int8_t *t = e->reff[0]; e->reff[0] = e->reff[1]; e->reff[1] = t; }
    e->have_ref++;
    return 0;
}

/* reconstructed (deblocked) picture of the most recent frame, cropped to width x height, I420 */
void x264o_encoder_get_recon(x264o_encoder *e, uint8_t *out)
{
    int w = e->cfg.width, h = e->cfg.height, slot = (e->cur + e->slots - 1) % e->slots;
    const pixel *Y = luma_plane(e, slot, 0), *UV = chroma_plane(e, slot);
    for (int y = 0; y < h; y++) memcpy(out + (size_t)y * w, Y + (size_t)y * e->rs, w);
    uint8_t *u = out + (size_t)w * h, *v = u + (size_t)(w / 2) * (h / 2);
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++) { u[y * (w / 2) + x] = UV[(size_t)y * e->rs + 2 * x]; v[y * (w / 2) + x] = UV[(size_t)y * e->rs + 2 * x + 1]; }
}

/* debugging taps for parity tests: raw pointers into the newest reference (after rotate: slot cur^1) */
const uint8_t *x264o_encoder_ref_plane(x264o_encoder *e, int k, int *stride, int *rows)
{
    *stride = e->rs;
    int last = (e->cur + e->slots - 1) % e->slots;
    if (k < 4) { *rows = e->ch + 2 * PAD; return e->luma[last] + k * e->plane_bytes; }
    *rows = e->ch / 2 + 2 * CPAD;
    return e->chroma[last];
}

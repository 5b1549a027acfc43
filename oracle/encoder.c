/* oracle/encoder.c — CPU restatement of the per-frame encode hot path (TEST INFRASTRUCTURE ONLY).
 *
 * Frame-level shell of the executable specification behind x264_encoder_encode() (reference call site codec.c:1693;
 * [x264-upstream] encoder/encoder.c x264_slice_write, common/{frame,deblock,mc}.c): ingest, per-macroblock quantisers, the
 * raster-order macroblock loop (analyse.c: x264_macroblock_analyse + x264_macroblock_encode in x264's own structure), QP_Y
 * inheritance, in-loop deblocking, half-pel planes, DPB rotation.  The HIP pipeline (x264vfw_amd/csrc/encoder.hip) must reproduce
 * its x264gpu_mb records, quantised levels and reconstructed frames bit-exactly.
 *
 * parity unpinned vs libx264 (not in /root/reference, see x264o.h): written from x264's published algorithm.
 */
#include "encoder_priv.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lambda * (2*log2(|mvd|+1) + 0.718 + (mvd != 0)) + 0.5, saturated to u16 (float math as in x264's x264_analyse_init_costs) */
void x264o_build_cost_mv(uint16_t *tab /* 2*MVCOST_HALF entries */, int lambda)
{
    for (int i = 0; i < MVCOST_HALF; i++) {
        const float logs = i ? log2f((float)(i + 1)) * 2.0f + 1.718f : 0.718f;
        int c = (int)((float)lambda * logs + 0.5f);
        if (c > 65535) c = 65535;
        tab[MVCOST_HALF + i] = (uint16_t)c;
        tab[MVCOST_HALF - i] = (uint16_t)c;
    }
    tab[0] = tab[1];
}

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
    e->slots = clampi(cfg->dpb > 0 ? cfg->dpb : cfg->refs, 1, X264O_MAX_SLOTS - 1) + 1;
    const size_t n = (size_t)e->mbw * e->mbh;
    for (int s = 0; s < e->slots; s++) {
        e->luma[s] = calloc(4, e->plane_bytes);
        e->chroma[s] = calloc(1, e->cplane_bytes);
        e->mv16[s] = calloc(n, sizeof(int16_t[2]));
        e->mbtype[s] = calloc(n, 1);
        e->colref[s] = calloc(n, 4); e->colmv[s] = calloc(n, sizeof(int16_t[4][2])); e->colref0[s] = calloc(n, 4);
    }
    for (int r = 1; r < X264O_MAX_REFS; r++) e->mvr[r] = calloc(n, sizeof(int16_t[2]));
    for (int r = 0; r < X264O_MAX_REFS; r++) e->mvr1[r] = calloc(n, sizeof(int16_t[2]));
    e->mbqp = malloc(n);
    e->amvd = calloc((size_t)n, 8); e->amvd1 = calloc((size_t)n, 8);
    x264o_quant_init(&e->qt, cfg->deadzone_inter, cfg->deadzone_intra);
    return e;
}

void x264o_encoder_destroy(x264o_encoder *e)
{
    if (!e) return;
    for (int s = 0; s < e->slots; s++) { free(e->luma[s]); free(e->chroma[s]); free(e->mv16[s]); free(e->mbtype[s]); free(e->colref[s]); free(e->colmv[s]); free(e->colref0[s]); }
    for (int r = 1; r < X264O_MAX_REFS; r++) free(e->mvr[r]);
    for (int r = 0; r < X264O_MAX_REFS; r++) free(e->mvr1[r]);
    for (int q = 0; q < 52; q++) free(e->cost_mv[q]);
    free(e->fenc_y); free(e->fenc_uv); free(e->mbqp); free(e->amvd); free(e->amvd1); free(e);
}

int x264o_encoder_mb_count(const x264o_encoder *e) { return e->mbw * e->mbh; }
void x264o_encoder_set_qp(x264o_encoder *e, int qp_i, int qp_p) { e->cfg.qp_i = qp_i; e->cfg.qp_p = qp_p; }
/* the float quantiser (rc->qpm) of x264o_encoder_encode's following pictures; 0 = the integer one */
void x264o_encoder_set_qpm(x264o_encoder *e, float qpm) { e->qpm_next = qpm; }
/* per-macroblock quantiser offsets (single floats) for the following pictures; the array must stay valid; NULL = back to the encoder's own AQ */
void x264o_encoder_set_mb_qp_offsets(x264o_encoder *e, const float *off) { e->ext_off = off; }
/* lookahead vectors of the NEXT picture against its predecessor ([nmb][2], lowres quarter-pels; first entry 0x7fff or NULL = none):
 * the extra 16x16 search candidate x264 takes from fenc->lowres_mvs[0][0] */
void x264o_encoder_set_lowres_mvs(x264o_encoder *e, const int16_t *mv) { e->lowres_mv = mv; }
void x264o_encoder_set_lowres_mvs1(x264o_encoder *e, const int16_t *mv) { e->lowres_mv1 = mv; }
/* tests: where to leave the predicted CAVLC bit count of every macroblock of the next pictures (NULL: off) */
void x264o_encoder_set_mb_bits_out(x264o_encoder *e, int *bits) { e->mb_bits = bits; }
/* tests: the CABAC context states (pStateIdx << 1 | valMPS) after the last slice coded — RD sessions with cabac only */
void x264o_encoder_cabac_states(const x264o_encoder *e, uint8_t *out) { memcpy(out, e->cabac_state, 460); }

const uint16_t *x264o_cost_mv_for(x264o_encoder *e, int qp)
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
 * energy = var(16x16 luma) + var(8x8 U) + var(8x8 V) with var = ssd - (sum^2 >> log2 n);
 * qp_adj = strength * (x264_log2(max(energy, 1)) - 14.427f) in single floats, strength = --aq-strength * 1.0397f (cfg.aq_strength);
 * x264_ratecontrol_mb_qp: the macroblock's quantiser = clip3((int)(rc->qpm + qp_adj + 0.5f)), qpm the picture's float quantiser. ---- */
#include "fixlut.h"
static void compute_mb_qp(x264o_encoder *e, int slice_qp, float qpm)
{
    const int n = e->mbw * e->mbh;
    if (qpm == 0.f) qpm = (float)slice_qp;          /* constant-quantiser sessions: rc->qpm = the slice's integer quantiser */
    if (e->ext_off) {        /* offsets decided by the lookahead (x264: frame->f_qp_offset / f_qp_offset_aq, read by x264_ratecontrol_mb_qp) */
        for (int i = 0; i < n; i++) e->mbqp[i] = (uint8_t)clampi(x264o_mb_qp(qpm, e->ext_off[i]), 1, 51);
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
            const float qp_adj = e->cfg.aq_strength * (x264o_log2(energy ? energy : 1) - 14.427f);
            e->mbqp[mby * e->mbw + mbx] = (uint8_t)clampi(x264o_mb_qp(qpm, qp_adj), 1, 51);
        }
}

/* QP_Y of a macroblock that sends no mb_qp_delta (no coefficients, not Intra16x16) is the previous macroblock's (7.4.5): the
 * deblocking filter and the next delta use that value (x264_macroblock_cache_save does the same) */
static void settle_mb_qp(x264o_encoder *e, x264gpu_mb *mbs, int slice_qp)
{
    int last = slice_qp;
    const int ns = e->cfg.slices > 1 ? e->cfg.slices : 1;
    for (int i = 0, sl = 0; i < e->mbw * e->mbh; i++) {
        x264gpu_mb *m = &mbs[i];
        if (i == ((e->mbh * sl + ns / 2) / ns) * e->mbw) { last = e->mbqp[i]; sl++; }      /* a slice starts from the slice quantiser = its first macroblock's */
        if (m->type != X264GPU_MB_I16x16 && !m->cbp_luma && !m->cbp_chroma) m->qp = (uint8_t)last;
        /* x264's entropy coders (qp_delta writers): an I16x16 with nothing coded at all (no DC either) does not spend a delta on RAISING
         * the quantiser — it takes the previous one, and that is the qp the loop filter then sees */
        else if (m->type == X264GPU_MB_I16x16 && !m->cbp_luma && !m->cbp_chroma && !((m->nnz >> 24) & 1) && m->qp > last) m->qp = (uint8_t)last;
        last = m->qp;
    }
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
static int is_b_inter(const x264gpu_mb *m) { return m->type >= X264GPU_MB_B_DIRECT && m->type <= X264GPU_MB_B_8x8; }

static int g_ref_pic[X264GPU_MAX_LIST + 1];       /* deblock_frame: picture (DPB slot) behind list-0 index r at [r + 1]; [0] = no reference */
static int edge_bs(const x264gpu_mb *p, int pbx, int pby, const x264gpu_mb *q, int qbx, int qby, int mb_edge)
{
    if (is_intra(p) || is_intra(q)) return mb_edge ? 4 : 3;
    if (blk_nnz(p, pbx, pby) || blk_nnz(q, qbx, qby)) return 2;
    int pi = (pby >> 1) * 2 + (pbx >> 1), qi = (qby >> 1) * 2 + (qbx >> 1);
    /* x264 compares reference PICTURES (deblock_ref_table): an index and its --weightp duplicate are the same picture */
    if (g_ref_pic[p->ref[pi] < 0 ? 0 : p->ref[pi] + 1] != g_ref_pic[q->ref[qi] < 0 ? 0 : q->ref[qi] + 1]) return 1;
    if (abs(p->mv[pi][0] - q->mv[qi][0]) >= 4 || abs(p->mv[pi][1] - q->mv[qi][1]) >= 4) return 1;
    /* B slices (deblock_strength_c with bframe): list 1 compared index by index as well — x264's lists never share a picture (list 0 holds
     * earlier, list 1 later pictures), so comparing per list equals the standard's comparison of picture sets */
    if (is_b_inter(p) && is_b_inter(q)) {
        if (p->ref1[pi] != q->ref1[qi]) return 1;
        const int px = p->ref1[pi] < 0 ? 0 : p->mv1[pi][0], py = p->ref1[pi] < 0 ? 0 : p->mv1[pi][1];
        const int qx = q->ref1[qi] < 0 ? 0 : q->mv1[qi][0], qy = q->ref1[qi] < 0 ? 0 : q->mv1[qi][1];
        if (abs(px - qx) >= 4 || abs(py - qy) >= 4) return 1;
    }
    return 0;
}

/* does a slice begin at macroblock row `mby` (x264 slice threads: rows split evenly, validate/threadslice arithmetic) */
static int slice_starts_at_row(const x264o_encoder *e, int mby)
{
    const int ns = e->cfg.slices > 1 ? e->cfg.slices : 1;
    for (int sl = 1; sl < ns; sl++) if ((e->mbh * sl + ns / 2) / ns == mby) return 1;
    return 0;
}

static void deblock_frame(x264o_encoder *e, const x264gpu_mb *mbs)
{
    int a_off = e->cfg.deblock_alpha * 2, b_off = e->cfg.deblock_beta * 2;   /* slice_alpha_c0_offset_div2 * 2 */
    pixel *Y = luma_plane(e, e->cur, 0), *UV = chroma_plane(e, e->cur);
    g_ref_pic[0] = -1;
    for (int r = 0; r < X264GPU_MAX_LIST; r++) g_ref_pic[r + 1] = e->slice_type == X264GPU_SLICE_P && r < e->nref_l[0] ? e->lslot[0][r] : 100 + r;      /* (B slices: list 0 holds no picture twice) */
    for (int mby = 0; mby < e->mbh; mby++)
        for (int mbx = 0; mbx < e->mbw; mbx++) {
            const x264gpu_mb *q = &mbs[mby * e->mbw + mbx];
            for (int dir = 0; dir < 2; dir++)          /* 0: vertical edges (filter across x), 1: horizontal */
                for (int edge = 0; edge < 4; edge++) {
                    const x264gpu_mb *p = q;
                    if ((edge & 1) && q->transform8x8) continue;     /* no transform edge at 4-sample offsets */
                    if (edge == 0) {
                        if (dir == 0) { if (mbx == 0) continue; p = &mbs[mby * e->mbw + mbx - 1]; }
                        else { if (mby == 0 || (!e->cfg.slices_plain && slice_starts_at_row(e, mby))) continue; p = &mbs[(mby - 1) * e->mbw + mbx]; }      /* slice threads (idc 2): not across slices; --slices N (idc 0): across */
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

/* h->mb.bipred_weight: implicit weights of 8.4.2.3.1 from the POC distances (x264_macroblock_bipred_init) */
static void bipred_init(x264o_encoder *e)
{
    for (int r0 = 0; r0 < e->nref_l[0]; r0++)
        for (int r1 = 0; r1 < e->nref_l[1]; r1++) {
            const int poc0 = e->slot_poc[e->lslot[0][r0]], poc1 = e->slot_poc[e->lslot[1][r1]];
            const int td = clampi(poc1 - poc0, -128, 127);
            int dsf = 256;
            if (td) { const int tb = clampi(e->poc - poc0, -128, 127), tx = (16384 + (abs(td) >> 1)) / td; dsf = clampi((tb * tx + 32) >> 6, -1024, 1023); }
            if (r1 == 0) e->dist_scale[r0] = dsf;          /* h->mb.dist_scale_factor[r0][0]: temporal direct scales the co-located vector by it */
            dsf >>= 2;
            e->bipred_weight[r0][r1] = (e->cfg.weightb && dsf >= -64 && dsf <= 128) ? 64 - dsf : 32;
        }
}

/* --direct auto: the skip-probe counts of the last B picture coded with direct_auto (h->stat.frame.i_direct_score: [0] temporal, [1] spatial) */
void x264o_encoder_direct_scores(const x264o_encoder *e, int out[2]) { out[0] = e->direct_score[0]; out[1] = e->direct_score[1]; }

/* One picture with explicit control (x264gpu_pic): slice type, quantiser, POC, destination slot, reference lists.  B pictures need cfg.rd
 * and cfg.cabac. */
int x264o_encoder_encode_pic(x264o_encoder *e, const uint8_t *i420, const x264gpu_pic *pic, x264gpu_mb *mbs, int16_t *levels)
{
    int slice_type = pic->slice_type;
    if (slice_type == X264GPU_SLICE_I_NONIDR) slice_type = X264GPU_SLICE_I;      /* same coding tools, the references stay */
    if (pic->dst < 0 || pic->dst >= e->slots) return -1;
    e->slice_type = slice_type;
    e->cur = pic->dst; e->poc = pic->poc; e->keep = pic->keep;
    for (int l = 0; l < 2; l++) {
        e->nref_l[l] = slice_type == X264GPU_SLICE_I ? 0 : l == 1 && slice_type != X264GPU_SLICE_B ? 0 : pic->nref[l];
        if (e->nref_l[l] > X264O_MAX_REFS) return -1;
        for (int r = 0; r < e->nref_l[l]; r++) { e->lslot[l][r] = pic->slot[l][r]; if (pic->slot[l][r] < 0 || pic->slot[l][r] >= e->slots || pic->slot[l][r] == pic->dst) return -1; }
    }
    if (slice_type != X264GPU_SLICE_I && !e->nref_l[0]) return -1;
    /* x264 slice init: h->mb.b_dct_decimate = B slice || (--dct-decimate && not an I slice): B slices decimate whatever the option says */
    const int cfg_decimate = e->cfg.dct_decimate;
    if (slice_type == X264GPU_SLICE_B) e->cfg.dct_decimate = 1;
    if (slice_type == X264GPU_SLICE_B && !e->nref_l[1]) return -1;
    e->nref = e->nref_l[0];
    e->blind_dupe = -1;
    for (int r = 0; r < X264GPU_MAX_LIST; r++) {
        const int on = slice_type == X264GPU_SLICE_P && r < e->nref && pic->wl0[r].on;
        e->wl0[r].on = on; e->wl0[r].denom = on ? pic->wl0[r].denom : 0; e->wl0[r].scale = on ? pic->wl0[r].scale : 1; e->wl0[r].offset = on ? pic->wl0[r].offset : 0;
        for (int c = 0; c < 2; c++) {
            const int con = slice_type == X264GPU_SLICE_P && r < e->nref && pic->wc0[r].on[c];
            e->wc0[r].on[c] = con; e->wc0[r].scale[c] = con ? pic->wc0[r].scale[c] : 1; e->wc0[r].offset[c] = con ? pic->wc0[r].offset[c] : 0;
        }
        e->wc0[r].denom = (e->wc0[r].on[0] || e->wc0[r].on[1]) ? pic->wc0[r].denom : 0;
    }
    if (slice_type == X264GPU_SLICE_P && pic->blind_dupe > 0 && pic->blind_dupe < e->nref) e->blind_dupe = pic->blind_dupe;
    e->direct_temporal = 0; e->direct_auto = 0; e->direct_score[0] = e->direct_score[1] = 0;
    if (slice_type == X264GPU_SLICE_B) {
        bipred_init(e);
        e->direct_temporal = pic->direct_temporal != 0; e->direct_auto = pic->direct_auto != 0;
        /* x264_macroblock_slice_init: map_col_to_list0[i] = the list-0 index of this picture that shows the picture behind index i of the co-located
         * picture's list 0 (-2: none) */
        const int cs = e->lslot[1][0];
        for (int i = 0; i < X264GPU_MAX_LIST; i++) {
            e->map_col_to_list0[i] = -2;
            if (i >= e->slot_nref[cs]) continue;
            for (int j = 0; j < e->nref_l[0]; j++) if (e->slot_poc[e->lslot[0][j]] == e->slot_l0poc[cs][i]) { e->map_col_to_list0[i] = j; break; }
        }
    }
    ingest(e, i420);
    const int slice_qp = pic->qp;
    if (pic->qpm != 0.f && !(pic->qpm > (float)slice_qp - 1.f && pic->qpm < (float)slice_qp + 1.f)) return -1;      /* qp is the rounding of qpm */
    compute_mb_qp(e, slice_qp, pic->qpm);
    e->mbs = mbs; e->levels = levels; e->intra_count = 0;
    e->slot_nref[e->cur] = e->nref; e->slot_poc[e->cur] = e->poc; e->slot_ref0poc[e->cur] = e->nref ? e->slot_poc[ref_slot(e, 0)] : 0;
    for (int r = 0; r < X264GPU_MAX_LIST; r++) e->slot_l0poc[e->cur][r] = r < e->nref ? e->slot_poc[ref_slot(e, r)] : 0;
    /* the macroblock loop: raster order, every macroblock analysed AND coded before the next one starts (x264_slice_write) */
    const int ns = e->cfg.slices > 1 ? e->cfg.slices : 1;
    for (int sl = 0; sl < ns; sl++) {
        /* x264 slice threads: rows split evenly; each thread starts with empty frame statistics (h->stat.frame).  Plain --slices N are coded one
         * after the other by one thread: the statistics (the fast-intra decision reads the intra count) run on through the picture */
        e->row0 = (e->mbh * sl + ns / 2) / ns; e->row1 = (e->mbh * (sl + 1) + ns / 2) / ns;
        if (!e->cfg.slices_plain) e->intra_count = 0;
        /* x264_slice_write: the slice's quantiser — header, CABAC context initialisation, start of the mb_qp_delta chain — is its FIRST macroblock's */
        const int sl_qp = e->mbqp[e->row0 * e->mbw];
        e->last_qp = sl_qp;
        if (e->cfg.cabac && (e->cfg.rd || e->cfg.trellis)) { x264o_cabac_init_states(e->cabac_state, slice_type != X264GPU_SLICE_I, sl_qp); e->last_dqp = 0; }
        for (int mby = e->row0; mby < e->row1; mby++)
            for (int mbx = 0; mbx < e->mbw; mbx++) {
                const int mi = mby * e->mbw + mbx;
                x264o_macroblock(e, mbx, mby);
                const x264gpu_mb *m = &mbs[mi];
                e->mbtype[e->cur][mi] = m->type;
                /* the motion a later B picture's direct prediction reads when this picture heads its list 1 */
                for (int k = 0; k < 4; k++) {
                    const int intra = m->type <= X264GPU_MB_I16x16, b = m->type >= X264GPU_MB_B_DIRECT;
                    const int use1 = !intra && b && m->ref[k] < 0;
                    e->colref[e->cur][mi][k] = (int8_t)(intra ? -1 : use1 ? m->ref1[k] : m->ref[k]);
                    e->colref0[e->cur][mi][k] = (int8_t)(intra ? -1 : m->ref[k]);
                    e->colmv[e->cur][mi][k][0] = intra ? 0 : use1 ? m->mv1[k][0] : m->mv[k][0];
                    e->colmv[e->cur][mi][k][1] = intra ? 0 : use1 ? m->mv1[k][1] : m->mv[k][1];
                }
            }
    }
    e->row0 = 0; e->row1 = e->mbh;
    if (e->cfg.aq_mode || e->ext_off) settle_mb_qp(e, mbs, slice_qp);
    if (e->cfg.deblock) deblock_frame(e, mbs);
    if (pic->keep) filter_frame(e);
    e->cfg.dct_decimate = cfg_decimate;
    return 0;
}

/* The I / P stream of the earlier rounds: sliding-window DPB, reference index r = the picture coded r + 1 pictures ago */
int x264o_encoder_encode(x264o_encoder *e, const uint8_t *i420, int slice_type, x264gpu_mb *mbs, int16_t *levels)
{
    if (slice_type == X264GPU_SLICE_P && !e->have_ref) return -1;
    if (slice_type == X264GPU_SLICE_B) return -1;
    if (slice_type == X264GPU_SLICE_I) { e->have_ref = 0; e->ring_poc = 0; }      /* IDR: the DPB is emptied */
    x264gpu_pic pic;
    memset(&pic, 0, sizeof(pic));
    const int ring = clampi(e->cfg.refs, 1, 5) + 1;
    pic.slice_type = slice_type; pic.qp = slice_type == X264GPU_SLICE_P ? e->cfg.qp_p : e->cfg.qp_i; pic.poc = e->ring_poc; pic.dst = e->ring_cur; pic.keep = 1; pic.blind_dupe = -1; pic.qpm = e->qpm_next;
    pic.nref[0] = slice_type == X264GPU_SLICE_P ? (e->have_ref < ring - 1 ? e->have_ref : ring - 1) : 0;
    for (int r = 0; r < pic.nref[0]; r++) pic.slot[0][r] = (int8_t)((e->ring_cur - 1 - r + 2 * ring) % ring);
    const int rc = x264o_encoder_encode_pic(e, i420, &pic, mbs, levels);
    if (rc) return rc;
    /* rotate: the frame just built becomes the reference */
    e->last_slot = e->ring_cur;
    e->ring_cur = (e->ring_cur + 1) % ring;
    e->have_ref++;
    e->ring_poc += 2;
    return 0;
}

/* reconstructed (deblocked) picture of the most recent frame, cropped to width x height, I420 */
void x264o_encoder_get_recon(x264o_encoder *e, uint8_t *out)
{
    int w = e->cfg.width, h = e->cfg.height, slot = e->cur;
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
    int last = e->cur;
    if (k < 4) { *rows = e->ch + 2 * PAD; return e->luma[last] + k * e->plane_bytes; }
    *rows = e->ch / 2 + 2 * CPAD;
    return e->chroma[last];
}

/* tests: the oracle's own derivation of the tables (fixlut.h), to be compared with the product's literals */
void x264o_fixed_point_luts(float log2_lut[128], uint16_t exp2_lut[64])
{
    memcpy(log2_lut, x264o_log2f_lut(), 128 * sizeof(float));
    memcpy(exp2_lut, x264o_exp2_lut(), 64 * sizeof(uint16_t));
}
/* ... and the float primitives themselves, for known-answer tests */
float x264o_log2_f(uint32_t x) { return x264o_log2(x); }
int x264o_exp2fix8_f(float x) { return x264o_exp2fix8(x); }
int x264o_mb_qp_f(float qpm, float off) { return x264o_mb_qp(qpm, off); }

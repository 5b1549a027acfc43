/* oracle/lookahead.c — CPU restatement of the lookahead frame cost (TEST INFRASTRUCTURE ONLY; the product never links it).
 *
 * Follows [x264-upstream] encoder/slicetype.c: x264_slicetype_frame_cost -> x264_slicetype_mb_cost for the one pair this
 * round's I/P pipeline needs (p0 = previous picture, b = p1 = the new picture), reached from the reference at codec.c:1693
 * (x264_encoder_encode -> lookahead -> slicetype decision / x264_rc_analyse_slice).  SURVEY.md §8a row A12, §8f row 2.
 *
 *   - half-resolution planes (x264_frame_init_lowres_core: the four half-sample phases) of the mod-16
 *     expanded luma, borders replicated;
 *   - per 8x8 half-resolution block: intra cost = min SATD over 8x8c DC/H/V(/P) and Intra_8x8 modes 3..8 on SOURCE neighbours
 *     (+ 5 lambda + lowres_penalty 4); inter cost = x264_me_search on the previous picture's half-resolution planes (hexagon,
 *     --merange, sub-pel as subme 4: one half-pel SAD + one quarter-pel SATD diamond), lambda of qp 12, minus the mvd-0 cost,
 *     + 5 lambda when the vector is not zero; a zero predictor with SATD < 64 at the zero vector skips the search;
 *   - frame sums over the blocks that count for the frame score (not on the picture border unless the picture is <= 2 blocks).
 *
 * parity unpinned vs libx264 (see x264o.h).  One deliberate difference, the same as in the frame pipeline (encoder.c): x264
 * predicts each block's vector from its already-searched right / lower neighbours (reverse raster order); here the
 * predictor is the median of the PREVIOUS picture's half-resolution field (left, top, top-right), which makes every block
 * independent on the GPU.
 */
#include "x264o.h"
#include <stdlib.h>
#include <string.h>

#define LPAD 32

int  x264o_lambda(int qp);
void x264o_build_cost_mv(uint16_t *tab, int lambda);

typedef struct x264o_lookahead {
    int w, h, bw, bh;            /* picture size; half-resolution blocks (= macroblocks) */
    int lw, lh, ls;              /* half-resolution plane size and stride */
    size_t lplane;               /* bytes of one padded plane */
    pixel *planes[2];            /* two pictures x four padded planes */
    int16_t (*mv[2])[2];         /* per block: quarter-sample vector found by the search */
    int8_t *inter[2];            /* per block: 1 = vector valid (0: no previous picture or intra won) */
    int cur, have_prev, me_range, subme, lambda;
    uint16_t *cost_mv;           /* 32768-centred table as in encoder.c */
    pixel *tmp;
} x264o_lookahead;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
static int median3(int a, int b, int c) { int mn = a < b ? a : b, mx = a < b ? b : a; return c < mn ? mn : c > mx ? mx : c; }
static pixel *la_plane(x264o_lookahead *la, int slot, int k) { return la->planes[slot] + k * la->lplane + (size_t)LPAD * la->ls + LPAD; }

x264o_lookahead *x264o_lookahead_create(int width, int height, int me_range, int subme)
{
    x264o_lookahead *la = calloc(1, sizeof(*la));
    la->w = width; la->h = height;
    la->bw = (width + 15) / 16; la->bh = (height + 15) / 16;
    la->lw = la->bw * 8; la->lh = la->bh * 8;
    la->ls = (la->lw + 2 * LPAD + 63) / 64 * 64;
    la->lplane = (size_t)la->ls * (la->lh + 2 * LPAD);
    for (int s = 0; s < 2; s++) {
        la->planes[s] = calloc(4, la->lplane);
        la->mv[s] = calloc((size_t)la->bw * la->bh, sizeof(int16_t[2]));
        la->inter[s] = calloc((size_t)la->bw * la->bh, 1);
    }
    la->me_range = clampi(me_range, 4, 16); la->subme = subme;
    la->lambda = x264o_lambda(12);                                   /* X264_LOOKAHEAD_QP */
    la->cost_mv = malloc(2 * 32768 * sizeof(uint16_t));
    x264o_build_cost_mv(la->cost_mv, la->lambda);
    la->tmp = malloc((size_t)la->bw * 16 * la->bh * 16);
    return la;
}

void x264o_lookahead_destroy(x264o_lookahead *la)
{
    if (!la) return;
    for (int s = 0; s < 2; s++) { free(la->planes[s]); free(la->mv[s]); free(la->inter[s]); }
    free(la->cost_mv); free(la->tmp); free(la);
}

static const int8_t hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
static const int8_t square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
static const int8_t mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };

/* x264_me_search_ref for one 8x8 half-resolution block: start candidates (predictor, zero, co-located) in order, hexagon +
 * square on SAD, one half-pel SAD diamond, SATD at the best, one quarter-pel SATD diamond (subme 4).  Same order and
 * tie-breaks as me_search_block in encoder.c. */
static int la_search(x264o_lookahead *la, int bx, int by, const int mvp[2], int has_col, const int col[2], int out_mv[2])
{
    const pixel *fenc = la_plane(la, la->cur, 0) + (size_t)by * 8 * la->ls + bx * 8;
    pixel *planes[4];
    for (int k = 0; k < 4; k++) planes[k] = la_plane(la, la->cur ^ 1, k);
    const pixel *full = planes[0] + (size_t)by * 8 * la->ls + bx * 8;
    const uint16_t *cm = la->cost_mv + 32768, *cmx = cm - mvp[0], *cmy = cm - mvp[1];
    const int ls = la->ls;
    int smin[2] = { 4 * (-8 * bx - 12), 4 * (-8 * by - 12) }, smax[2] = { 4 * (8 * (la->bw - bx - 1) + 12), 4 * (8 * (la->bh - by - 1) + 12) };
    int fmin[2], fmax[2];
    for (int k = 0; k < 2; k++) { fmin[k] = (smin[k] >> 2) + 6; fmax[k] = (smax[k] >> 2) - 6; }
#define FPEL_COST(mx, my) (x264o_sad(fenc, ls, full + (my) * ls + (mx), ls, 8, 8) + cmx[(mx) * 4] + cmy[(my) * 4])
    int cand[3][2] = { { (mvp[0] + 2) >> 2, (mvp[1] + 2) >> 2 }, { 0, 0 }, { (col[0] + 2) >> 2, (col[1] + 2) >> 2 } };
    int bmx = 0, bmy = 0, bcost = 1 << 28;
    for (int i = 0; i < (has_col ? 3 : 2); i++) {
        int cx = clampi(cand[i][0], fmin[0], fmax[0]), cy = clampi(cand[i][1], fmin[1], fmax[1]);
        int c = FPEL_COST(cx, cy);
        if (c < bcost) { bcost = c; bmx = cx; bmy = cy; }
    }
    {
        int key = bcost << 3;
        for (int k = 1; k <= 6; k++) {
            int c = (FPEL_COST(bmx + hex2[k][0], bmy + hex2[k][1]) << 3) + k + 1;
            if (c < key) key = c;
        }
        if (key & 7) {
            int dir = (key & 7) - 2;
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (int i = (la->me_range >> 1) - 1; i > 0 && bmx >= fmin[0] && bmx <= fmax[0] && bmy >= fmin[1] && bmy <= fmax[1]; i--) {
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
#undef FPEL_COST
    int mx = bmx * 4, my = bmy * 4;
    pixel pred[64];
    {   /* subme 4: one half-pel iteration (SAD), SATD at the best, one quarter-pel iteration (SATD) */
        static const int8_t dia[4][2] = { { 0, -2 }, { 0, 2 }, { -2, 0 }, { 2, 0 } }, qd[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        int omx = mx, omy = my;
        for (int k = 0; k < 4; k++) {
            int cx = omx + dia[k][0], cy = omy + dia[k][1];
            x264o_mc_luma(pred, 8, planes, ls, bx * 8, by * 8, cx, cy, 8, 8);
            int c = x264o_sad(fenc, ls, pred, 8, 8, 8) + cmx[cx] + cmy[cy];
            if (c < bcost) { bcost = c; mx = cx; my = cy; }
        }
        x264o_mc_luma(pred, 8, planes, ls, bx * 8, by * 8, mx, my, 8, 8);
        bcost = x264o_satd(fenc, ls, pred, 8, 8, 8) + cmx[mx] + cmy[my];
        if (!(my <= smin[1] || my >= smax[1] || mx <= smin[0] || mx >= smax[0])) {
            omx = mx; omy = my;
            for (int k = 0; k < 4; k++) {
                int cx = omx + qd[k][0], cy = omy + qd[k][1];
                x264o_mc_luma(pred, 8, planes, ls, bx * 8, by * 8, cx, cy, 8, 8);
                int c = x264o_satd(fenc, ls, pred, 8, 8, 8) + cmx[cx] + cmy[cy];
                if (c < bcost) { bcost = c; mx = cx; my = cy; }
            }
        }
    }
    out_mv[0] = mx; out_mv[1] = my;
    return bcost;
}

static int la_intra_cost(x264o_lookahead *la, int bx, int by)
{
    const pixel *src = la_plane(la, la->cur, 0) + (size_t)by * 8 * la->ls + bx * 8;
    pixel pred[64], edge[33];
    int best = 1 << 28;
    static const int cmodes[4] = { I_PRED_CHROMA_DC, I_PRED_CHROMA_H, I_PRED_CHROMA_V, I_PRED_CHROMA_P };
    for (int i = 0; i < (la->subme > 1 ? 4 : 3); i++) {
        x264o_predict_8x8c(pred, 8, src, la->ls, cmodes[i]);
        int c = x264o_satd(src, la->ls, pred, 8, 8, 8);
        if (c < best) best = c;
    }
    if (la->subme > 1) {
        x264o_predict_8x8_filter(src, la->ls, edge, X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPRIGHT | X264O_AVAIL_TOPLEFT);
        for (int m = I_PRED_4x4_DDL; m <= I_PRED_4x4_HU; m++) {
            x264o_predict_8x8(pred, 8, edge, m);
            int c = x264o_satd(src, la->ls, pred, 8, 8, 8);
            if (c < best) best = c;
        }
    }
    return best + 5 * la->lambda + 4;        /* intra_penalty + lowres_penalty */
}

/* out[0] = intra cost of the picture (i_cost_est[0][0]), out[1] = P cost against the previous picture (i_cost_est[1][0];
 * equals out[0] when there is none), out[2] = blocks of the frame score where intra won, out[3] = blocks in the frame score.
 * block_info (optional, bw*bh x 4 int32): per block intra cost, best cost, vector (x & 0xffff | y << 16, quarter-pel), 1 if inter won.
 * reset: forget the previous picture (IDR). */
int x264o_lookahead_frame_cost(x264o_lookahead *la, const uint8_t *i420, int reset, int32_t out[4], int32_t *block_info)
{
    const int cw = la->bw * 16, ch = la->bh * 16;
    if (reset) la->have_prev = 0;
    la->cur ^= 1;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) la->tmp[(size_t)y * cw + x] = i420[(size_t)clampi(y, 0, la->h - 1) * la->w + clampi(x, 0, la->w - 1)];
    pixel *dst[4];
    for (int k = 0; k < 4; k++) dst[k] = la_plane(la, la->cur, k);
    x264o_frame_init_lowres(la->tmp, cw, cw, ch, dst, la->ls);
    for (int k = 0; k < 4; k++)                                      /* x264_frame_expand_border_lowres */
        for (int y = -LPAD; y < la->lh + LPAD; y++)
            for (int x = -LPAD; x < la->lw + LPAD; x++)
                if (x < 0 || x >= la->lw || y < 0 || y >= la->lh)
                    dst[k][y * la->ls + x] = dst[k][clampi(y, 0, la->lh - 1) * la->ls + clampi(x, 0, la->lw - 1)];
    int64_t isum = 0, psum = 0;
    int nintra = 0, nscore = 0;
    int16_t (*pmv)[2] = la->mv[la->cur ^ 1], (*cmv)[2] = la->mv[la->cur];
    const int8_t *pin = la->inter[la->cur ^ 1];
    int8_t *cin = la->inter[la->cur];
    for (int by = 0; by < la->bh; by++)
        for (int bx = 0; bx < la->bw; bx++) {
            const int bi = by * la->bw + bx;
            const int score = (bx > 0 && bx < la->bw - 1 && by > 0 && by < la->bh - 1) || la->bw <= 2 || la->bh <= 2;
            const int icost = la_intra_cost(la, bx, by);
            int bcost = icost, intra = 1;
            cmv[bi][0] = cmv[bi][1] = 0; cin[bi] = 0;
            if (la->have_prev) {
                /* predictor: median of the previous picture's field (left, top, top-right / top-left), as encoder.c prev_mvp */
                int a[2] = { 0, 0 }, b[2] = { 0, 0 }, c[2] = { 0, 0 }, mvp[2], mv[2] = { 0, 0 }, pcost;
                const int ia = bx > 0, ib = by > 0, ic = by > 0 && bx + 1 < la->bw;
                if (ia && pin[bi - 1]) { a[0] = pmv[bi - 1][0]; a[1] = pmv[bi - 1][1]; }
                if (ib && pin[bi - la->bw]) { b[0] = pmv[bi - la->bw][0]; b[1] = pmv[bi - la->bw][1]; }
                if (ic) { if (pin[bi - la->bw + 1]) { c[0] = pmv[bi - la->bw + 1][0]; c[1] = pmv[bi - la->bw + 1][1]; } }
                else if (by > 0 && bx > 0 && pin[bi - la->bw - 1]) { c[0] = pmv[bi - la->bw - 1][0]; c[1] = pmv[bi - la->bw - 1][1]; }
                if (!ib && ia) { mvp[0] = a[0]; mvp[1] = a[1]; }
                else { mvp[0] = median3(a[0], b[0], c[0]); mvp[1] = median3(a[1], b[1], c[1]); }
                int skip = 0;
                if (!(mvp[0] | mvp[1])) {      /* fast skip of near-zero residual at the zero vector */
                    const pixel *fenc = la_plane(la, la->cur, 0) + (size_t)by * 8 * la->ls + bx * 8;
                    const pixel *ref0 = la_plane(la, la->cur ^ 1, 0) + (size_t)by * 8 * la->ls + bx * 8;
                    pcost = x264o_satd(fenc, la->ls, ref0, la->ls, 8, 8);
                    if (pcost < 64) skip = 1;
                }
                if (!skip) {
                    const int col[2] = { pmv[bi][0], pmv[bi][1] };
                    pcost = la_search(la, bx, by, mvp, pin[bi], col, mv);
                    pcost -= la->cost_mv[32768];                     /* "remove mvcost from skip mbs" */
                    if (mv[0] | mv[1]) pcost += 5 * la->lambda;
                }
                if (!(icost < pcost)) { bcost = pcost; intra = 0; cmv[bi][0] = (int16_t)mv[0]; cmv[bi][1] = (int16_t)mv[1]; cin[bi] = 1; }
            }
            if (block_info) { block_info[4 * bi] = icost; block_info[4 * bi + 1] = bcost; block_info[4 * bi + 2] = (cmv[bi][0] & 0xffff) | (int32_t)((uint32_t)cmv[bi][1] << 16); block_info[4 * bi + 3] = !intra; }
            if (score) { isum += icost; psum += bcost; nintra += intra && la->have_prev; nscore++; }
        }
    out[0] = (int32_t)isum; out[1] = (int32_t)psum; out[2] = nintra; out[3] = nscore;
    la->have_prev = 1;
    return 0;
}

/* ---- adaptive-quantisation offsets of a source picture, single floats (the lookahead computes them when the picture arrives, as
 * x264_adaptive_quant_frame does; same arithmetic as encoder.c compute_mb_qp, on the mod-16 expanded picture).  strength = aq-strength * 1.0397f ---- */
#include "fixlut.h"
#include <math.h>
static uint32_t la_ac_energy(const uint8_t *Y, const uint8_t *U, const uint8_t *V, int w, int h, int bx, int by)
{
    uint32_t sum = 0, sqr = 0, su = 0, squ = 0, sv = 0, sqv = 0;
    for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) { uint32_t p = Y[(size_t)clampi(by * 16 + r, 0, h - 1) * w + clampi(bx * 16 + c, 0, w - 1)]; sum += p; sqr += p * p; }
    for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) {
        size_t o = (size_t)clampi(by * 8 + r, 0, h / 2 - 1) * (w / 2) + clampi(bx * 8 + c, 0, w / 2 - 1);
        uint32_t u = U[o], v = V[o];
        su += u; squ += u * u; sv += v; sqv += v * v;
    }
    return (sqr - (sum * sum >> 8)) + (squ - (su * su >> 6)) + (sqv - (sv * sv >> 6));
}
void x264o_aq_offsets(const uint8_t *i420, int w, int h, float strength, float *out)
{
    const int bw = (w + 15) / 16, bh = (h + 15) / 16;
    const uint8_t *Y = i420, *U = i420 + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            const uint32_t energy = la_ac_energy(Y, U, V, w, h, bx, by);
            out[by * bw + bx] = strength * (x264o_log2(energy ? energy : 1) - 14.427f);
        }
}

/* ... --aq-mode 2 (auto-variance) and 3 (auto-variance with a bias to dark scenes), x264_adaptive_quant_frame's float path: every macroblock's
 * qp_adj = (energy + 1)^(1/8), the picture's mean and mean square of them, strength = aq-strength x mean,
 *   mode 2: strength x (qp_adj - avg)        mode 3: ... + aq-strength x (1 - 14 / qp_adj^2)        avg = mean - (mean square - 14) / (2 mean)
 * in single floats, summed in raster order as x264 does; the eighth root is three IEEE square roots (x264 calls powf: may differ in the last place).
 * strength = the plain aq-strength (without mode 1's 1.0397). */
void x264o_aq_offsets_mode(const uint8_t *i420, int w, int h, int mode, float aqs, float *out)
{
    const int bw = (w + 15) / 16, bh = (h + 15) / 16, nb = bw * bh;
    if (mode <= 1) { x264o_aq_offsets(i420, w, h, aqs, out); return; }
    const uint8_t *Y = i420, *U = i420 + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    float *adj = malloc((size_t)nb * sizeof(float));
    float avg_adj = 0.f, avg_adj_pow2 = 0.f;
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            const uint32_t energy = la_ac_energy(Y, U, V, w, h, bx, by);
            const float q = sqrtf(sqrtf(sqrtf((float)energy + 1.f)));
            adj[by * bw + bx] = q; avg_adj += q; avg_adj_pow2 += q * q;
        }
    avg_adj /= (float)nb; avg_adj_pow2 /= (float)nb;
    const float strength = aqs * avg_adj;
    avg_adj = avg_adj - 0.5f * (avg_adj_pow2 - 14.f) / avg_adj;
    for (int i = 0; i < nb; i++) {
        float q = strength * (adj[i] - avg_adj);
        if (mode == 3) q = q + aqs * (1.f - 14.f / (adj[i] * adj[i]));
        out[i] = q;
    }
    free(adj);
}

/* ---- macroblock-tree for an I/P-only stream ([x264-upstream] encoder/slicetype.c macroblock_tree, macroblock_tree_propagate,
 * mbtree_propagate_cost / _list of common/mc.c, macroblock_tree_finish), constant frame rate.  info[j] / aq[j]: per-block records
 * and AQ offsets of n consecutive pictures, j = 0 the one about to be coded.  Every picture hands the part of its cost that its
 * reference explains back to the blocks its vectors point at (bilinear split over four blocks, 15-bit saturating sums); the
 * oldest picture's blocks get  offset = aq - strength * (x264_log2(intra + propagated) - x264_log2(intra)).  x264's C expressions in single
 * floats: fps_factor = 1 / 512 (MBTREE_PRECISION 0.5f, constant frame rate), inverse quantiser scales = x264_exp2fix8 of the AQ offsets. ---- */
/* mbtree_propagate_cost (common/mc.c), one block */
static int la_propagate_amount(int propagate_in, int intra_cost, int inter_cost, int inv_qscale)
{
    if (!intra_cost) return 0;          /* (0 / 0 in x264: the conversion of the NaN is 0 in the stored int16) */
    const float fps = 1.f / 512.f;
    float propagate_intra = (float)(intra_cost * inv_qscale);
    float propagate_amount = (float)propagate_in + propagate_intra * fps;
    float propagate_num = (float)(intra_cost - inter_cost);
    float propagate_denom = (float)intra_cost;
    const int v = (int)(propagate_amount * propagate_num / propagate_denom + 0.5f);
    return v < 32767 ? v : 32767;
}
void x264o_mbtree(int bw, int bh, const int32_t *const *info, const float *const *aq, int n, float strength, float *out)
{
    const int nb = bw * bh;
    int32_t *prop = calloc((size_t)n * nb, sizeof(int32_t));           /* propagate cost of every picture's blocks */
    for (int j = n - 1; j >= 1; j--) {
        const int32_t *fi = info[j];
        int32_t *ref = prop + (size_t)(j - 1) * nb;
        for (int by = 0; by < bh; by++)
            for (int bx = 0; bx < bw; bx++) {
                const int i = by * bw + bx;
                const int intra = fi[4 * i] > 16383 ? 16383 : fi[4 * i], best = fi[4 * i + 1] > 16383 ? 16383 : fi[4 * i + 1];   /* LOWRES_COST_MASK */
                const int inter = best < intra ? best : intra, inv = x264o_exp2fix8(aq ? aq[j][i] : 0.f);
                const int amount = la_propagate_amount(prop[(size_t)j * nb + i], intra, inter, inv);
                if (!fi[4 * i + 3]) continue;                           /* intra block: nothing is explained by the reference */
                int x = (int16_t)(fi[4 * i + 2] & 0xffff), y = fi[4 * i + 2] >> 16;
#define CLIP_ADD(idx, v) do { int t_ = ref[idx] + (v); ref[idx] = t_ > 32767 ? 32767 : t_; } while (0)
                if (!(x | y)) { CLIP_ADD(i, amount); continue; }
                const int mbx = (x >> 5) + bx, mby = (y >> 5) + by;
                x &= 31; y &= 31;
                const int w0 = ((32 - y) * (32 - x) * amount + 512) >> 10, w1 = ((32 - y) * x * amount + 512) >> 10;
                const int w2 = (y * (32 - x) * amount + 512) >> 10, w3 = (y * x * amount + 512) >> 10;
                if (mby >= 0 && mby < bh) { if (mbx >= 0 && mbx < bw) CLIP_ADD(mby * bw + mbx, w0); if (mbx + 1 >= 0 && mbx + 1 < bw) CLIP_ADD(mby * bw + mbx + 1, w1); }
                if (mby + 1 >= 0 && mby + 1 < bh) { if (mbx >= 0 && mbx < bw) CLIP_ADD((mby + 1) * bw + mbx, w2); if (mbx + 1 >= 0 && mbx + 1 < bw) CLIP_ADD((mby + 1) * bw + mbx + 1, w3); }
#undef CLIP_ADD
            }
    }
    for (int i = 0; i < nb; i++) {
        const float a = aq ? aq[0][i] : 0.f;
        const int icost = info[0][4 * i] > 16383 ? 16383 : info[0][4 * i];
        const int intra = (icost * x264o_exp2fix8(a) + 128) >> 8;
        float off = a;
        if (intra) {
            const int p2 = prop[i] * 2;                                 /* (propagate * fps_factor + 128) >> 8, fps_factor = 512 */
            const float log2_ratio = x264o_log2((uint32_t)(intra + p2)) - x264o_log2((uint32_t)intra) + 0.f;
            off = a - strength * log2_ratio;
        }
        out[i] = off;
    }
    free(prop);
}

/* oracle/slicetype.c — CPU restatement of the lookahead's frame costs in x264's own structure (TEST INFRASTRUCTURE ONLY; the product
 * never links it).
 *
 * Follows [x264-upstream] encoder/slicetype.c: slicetype_frame_cost -> slicetype_slice_cost -> slicetype_mb_cost for ANY triple
 * (p0, p1, b) of pictures held in the lookahead — I costs (p0 = p1 = b), P costs (b = p1) and B costs (p0 < b < p1) —, reached from the
 * reference at codec.c:1693 (x264_encoder_encode -> x264_lookahead_get_frames -> x264_slicetype_decide / _analyse: scenecut, --b-adapt,
 * x264_rc_analyse_slice).  SURVEY.md §8a row A12.  Unlike oracle/lookahead.c (round 1's P-only cost with predictors from the previous
 * picture's field) this file keeps x264's dependencies:
 *   - macroblocks (8x8 blocks of the half-resolution planes) are visited in REVERSE raster order and a block's vector predictor is the
 *     median of its right / lower / lower-left / lower-right neighbours' vectors of the same search ("reverse-order MV prediction": the
 *     vectors are later handed to the main encoder as candidates);
 *   - vectors and costs of a (picture, list, distance) search are cached with the picture (lowres_mvs / lowres_mv_costs) and reused by
 *     every later triple that needs them (do_search); the intra cost of a picture is computed once (b_intra_calculated);
 *   - B costs: the scaled co-located vector pair of p1's search towards p0 and the zero pair as bidirectional candidates, both lists'
 *     searches, then the pair of the two search results (+ 5 lambda), implicit --weightb weight from the distances; no intra in B;
 *   - x264_me_search itself is the main encoder's (oracle/analyse.c me_search_ref through x264o_lowres_me_search) with the lookahead's
 *     settings (lowres_context_init: qp 12, me <= hex, sub-pel level 4).
 * Edge blocks are only visited when x264 would (mbtree / VBV sessions, or pictures of <= 2 blocks: do_edges); vectors of blocks never
 * visited read as zero here (x264 leaves those entries as allocated).
 * parity unpinned vs libx264 (see x264o.h): restated from memory of the upstream file.
 */
#include "encoder_priv.h"
#include <stdlib.h>
#include <string.h>

#define ST_MAX_B 16
#define LOWRES_COST_MASK ((1 << 14) - 1)
#define LOWRES_COST_SHIFT 14

int x264o_lowres_me_search(x264o_encoder *lo, int slot, int bx, int by, const int mvp[2], int (*mvc)[2], int i_mvc, const int lim[8], int param_subme, int mv[2], int *cost_mv);

typedef struct {
    int16_t (*mvs[2][ST_MAX_B + 1])[2];      /* lowres_mvs[list][distance - 1]; [0][0] == 0x7fff: not searched yet */
    int *mv_costs[2][ST_MAX_B + 1];
    int *intra_cost;
    uint16_t *lowres_costs[ST_MAX_B + 2][ST_MAX_B + 2];
    int cost_est[ST_MAX_B + 2][ST_MAX_B + 2];
    int intra_mbs[ST_MAX_B + 2];
    int intra_calculated;
} st_frame;

typedef struct x264o_slicetype {
    int w, h, bw, bh, nb, lw, lh;
    int nslots, bframes, param_subme, weightb, mv_range, do_edges, bframe_bias;
    x264o_encoder lo;            /* shell over the half-resolution plane sets (see x264o_lowres_me_search) */
    st_frame *fr;
    pixel *tmp;
    struct st_tree_s { int32_t *prop; float *aq; } *tree;        /* per slot: macroblock-tree propagate costs, AQ offsets (f_qp_offset_aq) */
} x264o_slicetype;

x264o_slicetype *x264o_slicetype_create(int width, int height, int slots, int bframes, int me_method, int subme, int me_range, int weightb, int mv_range, int do_edges)
{
    x264o_slicetype *st = calloc(1, sizeof(*st));
    st->w = width; st->h = height; st->bw = (width + 15) / 16; st->bh = (height + 15) / 16; st->nb = st->bw * st->bh;
    st->lw = st->bw * 8; st->lh = st->bh * 8;
    st->nslots = slots < X264O_MAX_SLOTS ? slots : X264O_MAX_SLOTS; st->bframes = bframes < ST_MAX_B ? bframes : ST_MAX_B;
    st->param_subme = subme; st->weightb = weightb; st->mv_range = mv_range > 0 ? mv_range : 512;
    st->do_edges = do_edges || st->bw <= 2 || st->bh <= 2;
    x264o_encoder *lo = &st->lo;
    lo->cfg.me_range = clampi(me_range, 4, 16);
    lo->cfg.me_method = subme > 1 ? (me_method < 1 ? me_method : 1) : 0;           /* min(hex, --me), or dia */
    lo->rs = (st->lw + 2 * PAD + 63) / 64 * 64; lo->fs = lo->rs;
    lo->plane_bytes = (size_t)lo->rs * (st->lh + 2 * PAD);
    for (int s = 0; s < st->nslots; s++) lo->luma[s] = calloc(4, lo->plane_bytes);
    st->fr = calloc((size_t)st->nslots, sizeof(st_frame));
    for (int s = 0; s < st->nslots; s++) {
        st_frame *f = &st->fr[s];
        for (int l = 0; l < 2; l++)
            for (int d = 0; d <= st->bframes; d++) { f->mvs[l][d] = calloc((size_t)st->nb, sizeof(int16_t[2])); f->mv_costs[l][d] = calloc((size_t)st->nb, sizeof(int)); }
        f->intra_cost = calloc((size_t)st->nb, sizeof(int));
        for (int i = 0; i <= st->bframes + 1; i++) for (int j = 0; j <= st->bframes + 1; j++) f->lowres_costs[i][j] = calloc((size_t)st->nb, sizeof(uint16_t));
    }
    st->tmp = malloc((size_t)st->bw * 16 * st->bh * 16);
    st->tree = calloc((size_t)st->nslots, sizeof(*st->tree));
    for (int s = 0; s < st->nslots; s++) { st->tree[s].prop = calloc((size_t)st->nb, sizeof(int32_t)); st->tree[s].aq = calloc((size_t)st->nb, sizeof(float)); }
    return st;
}

void x264o_slicetype_destroy(x264o_slicetype *st)
{
    if (!st) return;
    for (int s = 0; s < st->nslots; s++) {
        st_frame *f = &st->fr[s];
        free(st->lo.luma[s]);
        for (int l = 0; l < 2; l++) for (int d = 0; d <= st->bframes; d++) { free(f->mvs[l][d]); free(f->mv_costs[l][d]); }
        free(f->intra_cost);
        for (int i = 0; i <= st->bframes + 1; i++) for (int j = 0; j <= st->bframes + 1; j++) free(f->lowres_costs[i][j]);
    }
    for (int q = 0; q < 52; q++) free(st->lo.cost_mv[q]);
    for (int s = 0; s < st->nslots; s++) { free(st->tree[s].prop); free(st->tree[s].aq); }
    free(st->tree);
    free(st->fr); free(st->tmp); free(st);
}

/* x264_frame_init_lowres + x264_frame_expand_border_lowres of a new source picture, and the state x264_frame_pop_unused resets */
int x264o_slicetype_put_frame(x264o_slicetype *st, int slot, const uint8_t *i420)
{
    if (slot < 0 || slot >= st->nslots) return -1;
    const int cw = st->bw * 16, ch = st->bh * 16, ls = st->lo.rs;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) st->tmp[(size_t)y * cw + x] = i420[(size_t)clampi(y, 0, st->h - 1) * st->w + clampi(x, 0, st->w - 1)];
    pixel *dst[4];
    for (int k = 0; k < 4; k++) dst[k] = luma_plane(&st->lo, slot, k);
    x264o_frame_init_lowres(st->tmp, cw, cw, ch, dst, ls);
    for (int k = 0; k < 4; k++)
        for (int y = -PAD; y < st->lh + PAD; y++)
            for (int x = -PAD; x < st->lw + PAD; x++)
                if (x < 0 || x >= st->lw || y < 0 || y >= st->lh)
                    dst[k][y * ls + x] = dst[k][clampi(y, 0, st->lh - 1) * ls + clampi(x, 0, st->lw - 1)];
    st_frame *f = &st->fr[slot];
    for (int l = 0; l < 2; l++)
        for (int d = 0; d <= st->bframes; d++) { memset(f->mvs[l][d], 0, (size_t)st->nb * sizeof(int16_t[2])); f->mvs[l][d][0][0] = 0x7fff; memset(f->mv_costs[l][d], 0, (size_t)st->nb * sizeof(int)); }
    for (int i = 0; i <= st->bframes + 1; i++) for (int j = 0; j <= st->bframes + 1; j++) f->cost_est[i][j] = -1;
    memset(f->intra_mbs, 0, sizeof(f->intra_mbs));
    f->intra_calculated = 0;
    for (int i = 0; i < st->nb; i++) f->intra_cost[i] = 0xffff;          /* x264_frame_new: memset( i_intra_cost, -1 ) — blocks never costed never win a minimum */
    memset(st->tree[slot].prop, 0, (size_t)st->nb * sizeof(int32_t)); memset(st->tree[slot].aq, 0, (size_t)st->nb * sizeof(float));
    return 0;
}

typedef struct { int inter_cost_est, intra_cost_est, intra_mbs; } st_sums;

/* slicetype_mb_cost */
static void mb_cost(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int bx, int by, int dist_scale_factor, const int do_search[2], st_sums *out)
{
    x264o_encoder *lo = &st->lo;
    st_frame *fenc = &st->fr[sb], *fref1 = &st->fr[s1];
    const int b_bidir = d1 > 0, is_i = d0 == 0 && d1 == 0;
    const int bi = by * st->bw + bx, rs = lo->rs, lambda = x264o_lambda(12);
    const int bipred_weight = st->weightb ? 64 - (dist_scale_factor >> 2) : 32;
    const int score = (bx > 0 && bx < st->bw - 1 && by > 0 && by < st->bh - 1) || st->bw <= 2 || st->bh <= 2;
    const pixel *src = luma_plane(lo, sb, 0) + (size_t)by * 8 * rs + bx * 8;
    int i_bcost = COST_MAX, list_used = 0;
    const int lowres_penalty = 4;
    int mcost[2] = { COST_MAX, COST_MAX }, mmv[2][2] = { { 0, 0 }, { 0, 0 } };

    if (!is_i) {
        int lim[8];
        const int mvr = 2 * st->mv_range;
        lim[4] = 4 * (-8 * bx - 12) > -mvr ? 4 * (-8 * bx - 12) : -mvr;
        lim[5] = 4 * (8 * (st->bw - bx - 1) + 12) < mvr - 1 ? 4 * (8 * (st->bw - bx - 1) + 12) : mvr - 1;
        lim[6] = 4 * (-8 * by - 12) > -mvr ? 4 * (-8 * by - 12) : -mvr;
        lim[7] = 4 * (8 * (st->bh - by - 1) + 12) < mvr - 1 ? 4 * (8 * (st->bh - by - 1) + 12) : mvr - 1;
        lim[0] = lim[4] >> 2; lim[1] = lim[5] >> 2; lim[2] = lim[6] >> 2; lim[3] = lim[7] >> 2;
        pixel *pl[2][4];
        for (int k = 0; k < 4; k++) { pl[0][k] = luma_plane(lo, s0, k); pl[1][k] = luma_plane(lo, s1, k); }
        pixel p0[64], p1[64], avg[64];
#define TRY_BIDIR(mv0, mv1, penalty) do { \
        x264o_mc_luma(p0, 8, pl[0], rs, bx * 8, by * 8, (mv0)[0], (mv0)[1], 8, 8); \
        x264o_mc_luma(p1, 8, pl[1], rs, bx * 8, by * 8, (mv1)[0], (mv1)[1], 8, 8); \
        x264o_pixel_avg_weight(avg, 8, p0, 8, p1, 8, 8, 8, bipred_weight); \
        const int c_ = (penalty) * lambda + (st->param_subme > 1 ? x264o_satd(src, rs, avg, 8, 8, 8) : x264o_sad(src, rs, avg, 8, 8, 8)); \
        if (c_ < i_bcost) { i_bcost = c_; list_used = 3; } } while (0)
        if (b_bidir) {
            int dmv[2][2] = { { 0, 0 }, { 0, 0 } };
            int16_t (*rmv)[2] = fref1->mvs[0][d0 + d1 - 1];          /* fref1's search towards p0 */
            if (rmv[0][0] != 0x7fff) {
                dmv[0][0] = (rmv[bi][0] * dist_scale_factor + 128) >> 8; dmv[0][1] = (rmv[bi][1] * dist_scale_factor + 128) >> 8;
                dmv[1][0] = dmv[0][0] - rmv[bi][0]; dmv[1][1] = dmv[0][1] - rmv[bi][1];
                for (int l = 0; l < 2; l++) { dmv[l][0] = clampi(dmv[l][0], lim[4], lim[5]); dmv[l][1] = clampi(dmv[l][1], lim[6], lim[7]); }
                if (st->param_subme <= 1) for (int l = 0; l < 2; l++) { dmv[l][0] &= ~1; dmv[l][1] &= ~1; }
            }
            TRY_BIDIR(dmv[0], dmv[1], 0);
            if (dmv[0][0] | dmv[0][1] | dmv[1][0] | dmv[1][1]) {
                x264o_pixel_avg_weight(avg, 8, pl[0][0] + (size_t)by * 8 * rs + bx * 8, rs, pl[1][0] + (size_t)by * 8 * rs + bx * 8, rs, 8, 8, bipred_weight);
                const int c = st->param_subme > 1 ? x264o_satd(src, rs, avg, 8, 8, 8) : x264o_sad(src, rs, avg, 8, 8, 8);
                if (c < i_bcost) { i_bcost = c; list_used = 3; }
            }
        }
        for (int l = 0; l < 1 + b_bidir; l++) {
            int16_t (*fmv)[2] = fenc->mvs[l][(l ? d1 : d0) - 1];
            int *fcost = fenc->mv_costs[l][(l ? d1 : d0) - 1];
            if (do_search[l]) {
                /* reverse-order MV prediction: right, lower, lower-left, lower-right */
                int mvc[4][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 }, { 0, 0 } }, n = 0, mvp[2];
#define MVC(i) do { mvc[n][0] = fmv[i][0]; mvc[n][1] = fmv[i][1]; n++; } while (0)
                if (bx < st->bw - 1) MVC(bi + 1);
                if (by < st->bh - 1) {
                    MVC(bi + st->bw);
                    if (bx > 0) MVC(bi + st->bw - 1);
                    if (bx < st->bw - 1) MVC(bi + st->bw + 1);
                }
#undef MVC
                /* the "not searched yet" marker sits in entry 0 of the array: the frame-cost setup replaces it by 0 before the first block runs */
                if (n <= 1) { mvp[0] = mvc[0][0]; mvp[1] = mvc[0][1]; }
                else { mvp[0] = median3(mvc[0][0], mvc[1][0], mvc[2][0]); mvp[1] = median3(mvc[0][1], mvc[1][1], mvc[2][1]); }
                int skip = 0;
                if (!(mvp[0] | mvp[1])) {
                    const pixel *r0 = pl[l][0] + (size_t)by * 8 * rs + bx * 8;
                    mcost[l] = st->param_subme > 1 ? x264o_satd(src, rs, r0, rs, 8, 8) : x264o_sad(src, rs, r0, rs, 8, 8);
                    if (mcost[l] < 64) { mmv[l][0] = mmv[l][1] = 0; skip = 1; }
                }
                if (!skip) {
                    mcost[l] = x264o_lowres_me_search(lo, l ? s1 : s0, bx, by, mvp, mvc, n, lim, st->param_subme, mmv[l], NULL);
                    mcost[l] -= x264o_cost_mv_for(lo, 12)[0];                 /* remove mvcost from skip mbs */
                    if (mmv[l][0] | mmv[l][1]) mcost[l] += 5 * lambda;
                }
                fmv[bi][0] = (int16_t)mmv[l][0]; fmv[bi][1] = (int16_t)mmv[l][1];
                fcost[bi] = mcost[l];
            } else { mmv[l][0] = fmv[bi][0]; mmv[l][1] = fmv[bi][1]; mcost[l] = fcost[bi]; }
            if (mcost[l] < i_bcost) { i_bcost = mcost[l]; list_used = l + 1; }
        }
        if (b_bidir && (mmv[0][0] | mmv[0][1] | mmv[1][0] | mmv[1][1])) TRY_BIDIR(mmv[0], mmv[1], 5);
#undef TRY_BIDIR
    }
    /* lowres_intra_mb */
    if (!fenc->intra_calculated) {
        pixel pred[64], edge[33];
        int icost = COST_MAX;
        static const int cmodes[4] = { I_PRED_CHROMA_DC, I_PRED_CHROMA_H, I_PRED_CHROMA_V, I_PRED_CHROMA_P };
        for (int i = 0; i < (st->param_subme > 1 ? 4 : 3); i++) {
            x264o_predict_8x8c(pred, 8, src, rs, cmodes[i]);
            /* intra_mbcmp_x3_8x8c and, above subme 1, mbcmp of the plane prediction: SATD (SAD at subme <= 1) */
            const int c = st->param_subme > 1 ? x264o_satd(src, rs, pred, 8, 8, 8) : x264o_sad(src, rs, pred, 8, 8, 8);
            if (c < icost) icost = c;
        }
        if (st->param_subme > 1) {
            x264o_predict_8x8_filter(src, rs, edge, X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPRIGHT | X264O_AVAIL_TOPLEFT);
            for (int m = I_PRED_4x4_DDL; m <= I_PRED_4x4_HU; m++) {
                x264o_predict_8x8(pred, 8, edge, m);
                const int c = x264o_satd(src, rs, pred, 8, 8, 8);
                if (c < icost) icost = c;
            }
        }
        icost += 5 * lambda + lowres_penalty;
        fenc->intra_cost[bi] = icost;
        if (score) out->intra_cost_est += icost;
    }
    i_bcost += lowres_penalty;
    if (!b_bidir) {       /* intra blocks are not considered in B pictures */
        const int icost = fenc->intra_cost[bi], b_intra = icost < i_bcost;
        if (b_intra) { i_bcost = icost; list_used = 0; }
        if (score) out->intra_mbs += b_intra;
    }
    if (!is_i && score) out->inter_cost_est += i_bcost;
    fenc->lowres_costs[d0][d1][bi] = (uint16_t)((i_bcost < LOWRES_COST_MASK ? i_bcost : LOWRES_COST_MASK) + (list_used << LOWRES_COST_SHIFT));
}

int x264o_slicetype_frame_cost_w(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset);
/* --b-bias: h->param.i_bframe_bias in the scaling of B costs (score * 100 / (120 + bias)) */
void x264o_slicetype_set_bframe_bias(x264o_slicetype *st, int bias) { st->bframe_bias = bias < -90 ? -90 : bias > 100 ? 100 : bias; }
/* slicetype_frame_cost(p0, p1, b): slots of the three pictures and the distances d0 = b - p0, d1 = p1 - b.  Returns the frame's score. */
int x264o_slicetype_frame_cost(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1) { return x264o_slicetype_frame_cost_w(st, s0, s1, sb, d0, d1, 0, 1, 0, 0); }
/* ... with the explicit luma weight x264_weights_analyse( b_lookahead = 1 ) found for a P cost that is searched for the first time: the list-0 search
 * runs on the weighted reference (m[0].weight / fenc->weighted[0]); the zero-vector shortcut reads the unweighted plane, as x264's does */
int x264o_slicetype_frame_cost_w(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset)
{
    st->lo.wl0[0].on = on && d1 == 0 && d0 > 0; st->lo.wl0[0].scale = scale; st->lo.wl0[0].denom = denom; st->lo.wl0[0].offset = offset;
    if (d0 < 0 || d1 < 0 || d0 > st->bframes + 1 || d1 > st->bframes + 1 || (d1 > 0 && d0 == 0)) return -1;
    st_frame *fenc = &st->fr[sb];
    if (fenc->cost_est[d0][d1] >= 0) return fenc->cost_est[d0][d1];
    int do_search[2];
    do_search[0] = d0 > 0 && fenc->mvs[0][d0 - 1][0][0] == 0x7fff;
    do_search[1] = d1 > 0 && fenc->mvs[1][d1 - 1][0][0] == 0x7fff;
    if (do_search[0]) fenc->mvs[0][d0 - 1][0][0] = 0;
    if (do_search[1]) fenc->mvs[1][d1 - 1][0][0] = 0;
    int dsf = 128;
    if (d0 + d1 > 0 && d1 > 0) dsf = ((d0 << 8) + ((d0 + d1) >> 1)) / (d0 + d1);
    st_sums sums = { 0, 0, 0 };
    st->lo.fenc_y = luma_plane(&st->lo, sb, 0);
    const int e = st->do_edges;
    const int start_y = st->bh - 2 + e, end_y = 1 - e, start_x = st->bw - 2 + e, end_x = 1 - e;
    for (int by = start_y; by >= end_y; by--)
        for (int bx = start_x; bx >= end_x; bx--) mb_cost(st, s0, s1, sb, d0, d1, bx, by, dsf, do_search, &sums);
    if (d1 == 0) fenc->intra_mbs[d0] = sums.intra_mbs;
    if (!fenc->intra_calculated) fenc->cost_est[0][0] = sums.intra_cost_est;
    int score = d0 == 0 && d1 == 0 ? fenc->cost_est[0][0] : sums.inter_cost_est;
    if (d1 > 0) score = (int)((uint64_t)score * 100 / (120 + st->bframe_bias));
    else fenc->intra_calculated = 1;
    fenc->cost_est[d0][d1] = score;
    return score;
}

/* what the decisions read besides the score */
int x264o_slicetype_intra_mbs(const x264o_slicetype *st, int slot, int d0) { return st->fr[slot].intra_mbs[d0]; }
int x264o_slicetype_cost_est(const x264o_slicetype *st, int slot, int d0, int d1) { return st->fr[slot].cost_est[d0][d1]; }
const int16_t *x264o_slicetype_mvs(const x264o_slicetype *st, int slot, int list, int dist) { return &st->fr[slot].mvs[list][dist - 1][0][0]; }
const int *x264o_slicetype_mv_costs(const x264o_slicetype *st, int slot, int list, int dist) { return st->fr[slot].mv_costs[list][dist - 1]; }
const int *x264o_slicetype_intra_costs(const x264o_slicetype *st, int slot) { return st->fr[slot].intra_cost; }
const uint16_t *x264o_slicetype_lowres_costs(const x264o_slicetype *st, int slot, int d0, int d1) { return st->fr[slot].lowres_costs[d0][d1]; }

/* ---------------------------------------------------------------------------------------------------------------------------
 * macroblock-tree through B pictures ([x264-upstream] encoder/slicetype.c macroblock_tree_propagate, common/mc.c mbtree_propagate_cost /
 * mbtree_propagate_list, macroblock_tree_finish), constant frame rate.  Every picture hands the part of its cost that its references explain
 * back to the blocks its vectors point at: list 0 and list 1 as lowres_costs' list_used bits say, bi-predicted blocks split by the implicit
 * weight, bilinear split over four blocks, 15-bit saturating sums.  x264's C expressions in single floats as in oracle/lookahead.c
 * x264o_mbtree: fps_factor = 1 / 512 (MBTREE_PRECISION 0.5f at constant frame rate), inverse quantiser scales are x264_exp2fix8 of the AQ offsets. */
#include "fixlut.h"
static int st_inv_qscale(float aq) { return x264o_exp2fix8(aq); }
/* mbtree_propagate_cost (common/mc.c), one block */
static int st_propagate_amount(int propagate_in, int intra_cost, int inter_cost, int inv_qscale)
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
typedef struct st_tree_s st_tree;
static st_tree *st_tree_of(x264o_slicetype *st) { return st->tree; }
/* x264_adaptive_quant_frame's offsets of the picture in `slot` (f_qp_offset_aq; NULL = none): i_inv_qscale_factor follows from them */
void x264o_slicetype_set_aq(x264o_slicetype *st, int slot, const float *aq)
{
    st_tree *t = st_tree_of(st);
    if (aq) memcpy(t[slot].aq, aq, (size_t)st->nb * sizeof(float)); else memset(t[slot].aq, 0, (size_t)st->nb * sizeof(float));
}
/* fenc->i_cost_est_aq[b - p0][p1 - b]: the costs of the triple weighted block by block with the inverse quantiser scale of the picture's AQ offsets
 * (x264 slicetype_mb_cost: i_mb_cost_aq = (cost * i_inv_qscale_factor + 128) >> 8, summed over the blocks that count for the frame score) — what
 * x264_rc_analyse_slice hands the rate control in AQ sessions without macroblock-tree.  (The stored costs are capped at LOWRES_COST_MASK.) */
int x264o_slicetype_cost_aq(x264o_slicetype *st, int slot, int d0, int d1)
{
    st_frame *f = &st->fr[slot];
    if (d0 < 0 || d1 < 0 || d0 > st->bframes + 1 || d1 > st->bframes + 1 || f->cost_est[d0][d1] < 0) return -1;
    const float *aq = st_tree_of(st)[slot].aq;
    const int is_i = d0 == 0 && d1 == 0, e = st->do_edges;
    int sum = 0;
    for (int by = 0; by < st->bh; by++)
        for (int bx = 0; bx < st->bw; bx++) {
            const int score = (bx > 0 && bx < st->bw - 1 && by > 0 && by < st->bh - 1) || st->bw <= 2 || st->bh <= 2;
            if (!score) continue;
            (void)e;
            const int i = by * st->bw + bx;
            const int c = is_i ? f->intra_cost[i] : f->lowres_costs[d0][d1][i] & LOWRES_COST_MASK;
            sum += (c * st_inv_qscale(aq[i]) + 128) >> 8;
        }
    return sum;
}
void x264o_slicetype_clear_propagate(x264o_slicetype *st, int slot) { memset(st_tree_of(st)[slot].prop, 0, (size_t)st->nb * sizeof(int32_t)); }
/* macroblock_tree_propagate(p0, p1, b, referenced): the costs of (p0, p1, b) must have been computed */
int x264o_slicetype_propagate(x264o_slicetype *st, int s0, int s1, int sb, int d0, int d1, int referenced)
{
    st_tree *t = st_tree_of(st);
    st_frame *f = &st->fr[sb];
    if (d0 < 0 || d1 < 0 || d0 + d1 == 0 || f->cost_est[d0][d1] < 0) return -1;
    const int bw = st->bw, bh = st->bh;
    const int dsf = d1 > 0 ? ((d0 << 8) + ((d0 + d1) >> 1)) / (d0 + d1) : 256;
    const int bipw = st->weightb && d1 > 0 ? 64 - (dsf >> 2) : 32, bw2[2] = { bipw, 64 - bipw };
    const uint16_t *lc = f->lowres_costs[d0][d1];
    int32_t *refs[2] = { t[s0].prop, t[s1].prop };
    for (int by = 0; by < bh; by++)
        for (int bx = 0; bx < bw; bx++) {
            const int i = by * bw + bx;
            const int intra = f->intra_cost[i] > LOWRES_COST_MASK ? LOWRES_COST_MASK : f->intra_cost[i];
            const int best = lc[i] & LOWRES_COST_MASK, inter = best < intra ? best : intra, inv = st_inv_qscale(t[sb].aq[i]);
            const int amount = st_propagate_amount(referenced ? (t[sb].prop[i] > 32767 ? 32767 : t[sb].prop[i]) : 0, intra, inter, inv);
            const int used = lc[i] >> LOWRES_COST_SHIFT;
            for (int l = 0; l < (d1 > 0 ? 2 : 1); l++) {
                if (!(used & (1 << l))) continue;
                int la = amount;
                if (used == 3) la = (la * bw2[l] + 32) >> 6;
                const int16_t (*mv)[2] = f->mvs[l][(l ? d1 : d0) - 1];
                int x = mv[i][0], y = mv[i][1];
                int32_t *ref = refs[l];
#define CLIP_ADD(idx, v) do { int t_ = ref[idx] + (v); ref[idx] = t_ > 32767 ? 32767 : t_; } while (0)
                if (!(x | y)) { CLIP_ADD(i, la); continue; }
                const int mbx = (x >> 5) + bx, mby = (y >> 5) + by;
                x &= 31; y &= 31;
                const int w0 = ((32 - y) * (32 - x) * la + 512) >> 10, w1 = ((32 - y) * x * la + 512) >> 10;
                const int w2 = (y * (32 - x) * la + 512) >> 10, w3 = (y * x * la + 512) >> 10;
                if (mby >= 0 && mby < bh) { if (mbx >= 0 && mbx < bw) CLIP_ADD(mby * bw + mbx, w0); if (mbx + 1 >= 0 && mbx + 1 < bw) CLIP_ADD(mby * bw + mbx + 1, w1); }
                if (mby + 1 >= 0 && mby + 1 < bh) { if (mbx >= 0 && mbx < bw) CLIP_ADD((mby + 1) * bw + mbx, w2); if (mbx + 1 >= 0 && mbx + 1 < bw) CLIP_ADD((mby + 1) * bw + mbx + 1, w3); }
#undef CLIP_ADD
            }
        }
    return 0;
}
/* macroblock_tree_finish: out = f_qp_offset_aq - strength * (x264_log2(intra + propagated) - x264_log2(intra) + weightdelta), strength = 5.0f * (1.0f - qcomp),
 * weightdelta = 1 - f_weighted_cost_delta[ref0_distance - 1] where the lookahead's weight analysis found a weight for that distance (else 0) */
int x264o_slicetype_finish(x264o_slicetype *st, int slot, float strength, float weightdelta, float *out)
{
    st_tree *t = st_tree_of(st);
    const st_frame *f = &st->fr[slot];
    if (f->cost_est[0][0] < 0) return -1;          /* the intra costs exist once any cost of the picture has been computed */
    for (int i = 0; i < st->nb; i++) {
        const float a = t[slot].aq[i];
        const int icost = f->intra_cost[i] > LOWRES_COST_MASK ? LOWRES_COST_MASK : f->intra_cost[i];
        const int intra = (icost * st_inv_qscale(a) + 128) >> 8;
        float off = a;
        if (intra) {
            const int p2 = (t[slot].prop[i] > 32767 ? 32767 : t[slot].prop[i]) * 2;          /* (propagate * fps_factor + 128) >> 8, fps_factor = 512 */
            const float log2_ratio = x264o_log2((uint32_t)(intra + p2)) - x264o_log2((uint32_t)intra) + weightdelta;
            off = a - strength * log2_ratio;
        }
        out[i] = off;
    }
    return 0;
}
const int32_t *x264o_slicetype_propagate_cost(x264o_slicetype *st, int slot) { return st_tree_of(st)[slot].prop; }

/* ---------------------------------------------------------------------------------------------------------------------------
 * The primitives behind x264_weights_analyse ([x264-upstream] encoder/slicetype.c; --weightp): the pixel statistics of a source picture
 * (x264_adaptive_quant_frame: i_pixel_sum / i_pixel_ssd of the mod-16 expanded luma) and weight_cost_luma — the cost of predicting the
 * half-resolution picture from a reference under an explicit luma weight: per 8x8 block min(mbcmp(weighted reference block, source block),
 * intra cost), every block of the picture, + the bits the weights cost in the slice header.  The reference is motion-compensated by the
 * lookahead's vectors of (list 0, dist) when that search has run (weight_cost_init_luma), else taken in place.  The analysis itself — guessing
 * scale and offset from the statistics, the candidates around the guess, the 0.2 % gain threshold — is the caller's (host/encoder.cpp). */
void x264o_slicetype_pixel_stats(x264o_slicetype *st, int slot, const uint8_t *i420, uint64_t out[2])
{
    (void)slot;
    const int cw = st->bw * 16, ch = st->bh * 16;
    uint64_t sum = 0, sqr = 0;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) { const uint32_t p = i420[(size_t)clampi(y, 0, st->h - 1) * st->w + clampi(x, 0, st->w - 1)]; sum += p; sqr += p * p; }
    const uint64_t n = (uint64_t)cw * ch;
    out[0] = sum; out[1] = sqr - (sum * sum + n / 2) / n;
}
/* ... of the chroma planes: out = { sum Cb, ssd Cb, sum Cr, ssd Cr } (i_pixel_sum / i_pixel_ssd[1..2]) over the mod-16 expanded picture's chroma */
void x264o_slicetype_chroma_stats(x264o_slicetype *st, int slot, const uint8_t *i420, uint64_t out[4])
{
    (void)slot;
    const int cw = st->bw * 8, ch = st->bh * 8, pw = st->w / 2, ph = st->h / 2;
    for (int c = 0; c < 2; c++) {
        const uint8_t *pl = i420 + (size_t)st->w * st->h + (size_t)c * pw * ph;
        uint64_t sum = 0, sqr = 0;
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) { const uint32_t p = pl[(size_t)clampi(y, 0, ph - 1) * pw + clampi(x, 0, pw - 1)]; sum += p; sqr += p * p; }
        const uint64_t n = (uint64_t)cw * ch;
        out[2 * c] = sum; out[2 * c + 1] = sqr - (sum * sum + n / 2) / n;
    }
}
/* weight_cost_chroma of plane 1 (Cb) / 2 (Cr): the full-resolution chroma plane of the SOURCE picture against the reference's, the reference
 * motion-compensated 8x8 block by 8x8 block with the lookahead's vectors of (sf, list 0, dist) when that search has run (weight_cost_init_chroma:
 * mc_chroma with the half-resolution quarter-sample vector as it is), weighted; per block |sum of differences| (pixf.asd8: "the DC coefficient is
 * by far the most important part of the coding cost"), + the slice-header bits at four times luma's lambda.  i420_*: the raw pictures */
long x264o_slicetype_weight_cost_chroma(x264o_slicetype *st, int sf, const uint8_t *i420_fenc, const uint8_t *i420_ref, int dist, int plane, int on, int scale, int denom, int offset)
{
    st_frame *f = &st->fr[sf];
    const int pw = st->w / 2, ph = st->h / 2;
    const uint8_t *src = i420_fenc + (size_t)st->w * st->h + (size_t)(plane - 1) * pw * ph;
    const uint8_t *ref = i420_ref + (size_t)st->w * st->h + (size_t)(plane - 1) * pw * ph;
    const int16_t (*mv)[2] = dist > 0 && dist <= st->bframes + 1 && f->mvs[0][dist - 1][0][0] != 0x7fff ? f->mvs[0][dist - 1] : NULL;
    long cost = 0;
    for (int by = 0; by < st->bh; by++)
        for (int bx = 0; bx < st->bw; bx++) {
            const int i = by * st->bw + bx, mvx = mv ? mv[i][0] : 0, mvy = mv ? mv[i][1] : 0;
            const int d8x = mvx & 7, d8y = mvy & 7, cA = (8 - d8x) * (8 - d8y), cB = d8x * (8 - d8y), cC = (8 - d8x) * d8y, cD = d8x * d8y;
            int sum = 0;
            for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) {
                    const int px = bx * 8 + x + (mvx >> 3), py = by * 8 + y + (mvy >> 3);
                    const int x0 = clampi(px, 0, pw - 1), x1 = clampi(px + 1, 0, pw - 1), y0 = clampi(py, 0, ph - 1), y1 = clampi(py + 1, 0, ph - 1);
                    int p = (cA * ref[(size_t)y0 * pw + x0] + cB * ref[(size_t)y0 * pw + x1] + cC * ref[(size_t)y1 * pw + x0] + cD * ref[(size_t)y1 * pw + x1] + 32) >> 6;
                    if (on) { p = denom >= 1 ? ((p * scale + (1 << (denom - 1))) >> denom) + offset : p * scale + offset; p = p < 0 ? 0 : p > 255 ? 255 : p; }
                    sum += p - src[(size_t)clampi(by * 8 + y, 0, ph - 1) * pw + clampi(bx * 8 + x, 0, pw - 1)];
                }
            cost += sum < 0 ? -sum : sum;
        }
    if (on) {       /* weight_slice_header_cost, chroma: 4 x lambda(12) x (10 + denom bits (shared by the two planes) + 2 x (scale bits + offset bits)) */
        const int se_scale = bs_size_ue(scale > 0 ? 2 * scale - 1 : -2 * scale), se_off = bs_size_ue(offset > 0 ? 2 * offset - 1 : -2 * offset);
        cost += 4 * x264o_lambda(12) * (10 + bs_size_ue(denom) + 2 * (se_scale + se_off));
    }
    return cost;
}
/* dist: > 0 and that list-0 search of `sf` has run: motion-compensated reference; else the reference in place.  on = 0: unweighted */
long x264o_slicetype_weight_cost(x264o_slicetype *st, int sf, int sr, int dist, int on, int scale, int denom, int offset)
{
    x264o_encoder *lo = &st->lo;
    st_frame *f = &st->fr[sf];
    if (f->cost_est[0][0] < 0) return -1;
    const int rs = lo->rs;
    pixel *pl[4];
    for (int k = 0; k < 4; k++) pl[k] = luma_plane(lo, sr, k);
    const pixel *src = luma_plane(lo, sf, 0);
    const int16_t (*mv)[2] = dist > 0 && dist <= st->bframes + 1 && f->mvs[0][dist - 1][0][0] != 0x7fff ? f->mvs[0][dist - 1] : NULL;
    long cost = 0;
    for (int by = 0; by < st->bh; by++)
        for (int bx = 0; bx < st->bw; bx++) {
            const int i = by * st->bw + bx;
            pixel buf[64];
            x264o_mc_luma(buf, 8, pl, rs, bx * 8, by * 8, mv ? mv[i][0] : 0, mv ? mv[i][1] : 0, 8, 8);
            if (on) x264o_mc_weight(buf, 8, buf, 8, 8, 8, scale, denom, offset);
            const pixel *s = src + (size_t)by * 8 * rs + bx * 8;
            const int cmp = st->param_subme > 1 ? x264o_satd(buf, 8, s, rs, 8, 8) : x264o_sad(buf, 8, s, rs, 8, 8);
            cost += cmp < f->intra_cost[i] ? cmp : f->intra_cost[i];
        }
    if (on) {       /* weight_slice_header_cost: lambda(12) x slices x (10 + 2 x denom bits + 2 x (scale bits + offset bits)) */
        const int se_scale = bs_size_ue(scale > 0 ? 2 * scale - 1 : -2 * scale), se_off = bs_size_ue(offset > 0 ? 2 * offset - 1 : -2 * offset);
        cost += x264o_lambda(12) * (10 + bs_size_ue(denom) * 2 + 2 * (se_scale + se_off));
    }
    return cost;
}

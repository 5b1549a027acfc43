/* oracle/analyse.c — one macroblock of a slice: analysis and encode in x264's own structure (TEST INFRASTRUCTURE ONLY; see x264o.h).
 *
 * Restates, for I and P slices without RD (subme <= 5), the per-macroblock path behind x264_encoder_encode() (reference call site
 * codec.c:1693): [x264-upstream] encoder/analyse.c x264_macroblock_analyse (x264_mb_analyse_init, _inter_p16x16, _inter_p8x8,
 * _inter_p8x8_mixed_ref, _inter_p16x8, _inter_p8x16, _intra, _intra_chroma, _transform), encoder/me.c x264_me_search_ref /
 * refine_subpel / x264_me_refine_qpel, common/mvpred.c x264_mb_predict_mv*, encoder/macroblock.c x264_macroblock_encode,
 * x264_macroblock_probe_pskip, x264_mb_encode_i16x16 / _i4x4 / _i8x8 / _chroma.  libx264 is not in /root/reference: written from
 * the published algorithm, "parity unpinned" (x264o.h).
 *
 * Everything a macroblock reads comes from macroblocks BEFORE it in raster order of the same slice — motion vector predictors and
 * search candidates from the coded neighbours, intra prediction from their reconstruction, the fast-intra heuristic from the count
 * of intra macroblocks so far — so the loop is raster-serial, as x264's is.
 */
#include "encoder_priv.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { D_16x16 = 0, D_16x8 = 1, D_8x16 = 2, D_8x8 = 3 };

static const uint8_t blk_x[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
static const uint8_t blk_y[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
static const uint8_t idx_of[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };   /* [by][bx] */

typedef struct {          /* x264_me_t */
    int w, h, ox, oy;     /* block inside the macroblock */
    int list, ref, ref_cost;
    int mvp[2], mv[2];
    int cost, cost_mv;
} me_t;

typedef struct { int ref; int mv[2]; } nb_t;      /* ref: -2 unavailable, -1 intra, >= 0 reference index */

typedef struct {          /* x264_mb_analysis_t + the parts of h->mb the analysis uses */
    x264o_encoder *e;
    int mbx, mby, mi;
    int qp, qpc, lambda;
    const uint16_t *cost_mv;
    int mv_min[2], mv_max[2], smin[2], smax[2], fmin[2], fmax[2];
    int nref, subme, satd;                      /* satd: mbcmp is SATD (subme > 1), else SAD */
    int chroma_me;
    int type_left, type_top, type_tl, type_tr;  /* neighbour macroblock types, -1 = unavailable */
    int partition;                              /* h->mb.i_partition: selects the directional predictor rules */
    int cur_valid; nb_t cur8[4];                /* motion of this macroblock's 8x8 blocks decided so far (h->mb.cache), list 0 */
    int cur_valid1; nb_t cur8b[4];              /* ... list 1 (B slices) */
    int pskip_mv[2];
    int b_fast_intra, b_try_skip, b_early_terminate;
    me_t me16, me8[4], me16x8[2], me8x16[2];
    int mvc[X264O_MAX_REFS][5][2];              /* a->l0.mvc[ref][0] = 16x16 vector, [1..4] = 8x8 vectors */
    int cost8x8, cost16x8, cost8x16;
    int satd8x8[4], cost_est16x8_1, cost_est8x16_1;
    int satd_i16, satd_i8, satd_i4, satd_chroma;
    int pred16, pred8[4], pred4[16], predc;
    /* RD mode decision (cfg.rd): x264_mb_analysis_t i_mbrd, l0.i_rd16x16, forced transform size of the candidate being costed */
    int mbrd, rd16x16, force_t8, lambda2, chroma_lambda2_offset;
    /* B slices (x264_mb_analysis_t l0 / l1 and the bi / direct costs) */
    int nref_l[2];
    me_t me16l[2], bi16[2], me8l[2][4], me16x8l[2][2], me8x16l[2][2];
    int mvcl[2][X264O_MAX_REFS][5][2];
    int direct_ref[2][4], direct_mv[2][4][2], b_direct_available;      /* per 8x8 block (temporal direct: the blocks' list-0 indices differ) */
    int cost16x16bi, cost16x16direct, cost8x8direct[4], cost8x8bi, cost16x8bi, cost8x16bi;
    int satd8x8b[3][4], cost_est16x8[2], cost_est8x16[2];
    int sub8[4], part16x8[2], part8x16[2];      /* per block / half: 0 list 0, 1 list 1, 2 both, 3 direct (8x8 only) */
    int rd16l[2], rd16bi, rd16direct, rd8x8bi, rd16x8bi, rd8x16bi, bskip_cost;
    /* RD refinement (i_mbrd >= 2, subme >= 8): x264's h->mb.cache.non_zero_count of this macroblock as the LAST encode of any kind left it
     * (0 / 1 flags under CABAC: luma blocks 0..15, chroma AC plane * 4 + block), the by-products of the intra analysis that intra_rd_refine
     * reads (i_satd_i16x16_dir / i_satd_i8x8_dir / i_satd_chroma_dir, i_cbp_i8x8_luma), the chroma coded block pattern the refinement settled on */
    uint8_t nnzc[24];
    int satd_i16_dir[7], satd_i8_dir[4][12], satd_chroma_dir[7], cbp_i8;
} actx;

/* ---------------------------------------------------------------------------------------------------------------------------
 * lambda tables (x264_lambda_tab / x264_lambda2_tab: round(2^(qp/6-2)), round(0.9 * 2^((qp-12)/3) * 256)) */
static const uint8_t lambda_tab[52] = { 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14,
                                        16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91 };
static const int lambda2_tab[52] = { 14, 18, 22, 28, 36, 45, 57, 72, 91, 115, 145, 182, 230, 290, 365, 460, 580, 731, 921, 1160, 1462, 1843, 2322, 2926,
                                     3686, 4644, 5852, 7373, 9289, 11703, 14745, 18578, 23407, 29491, 37156, 46814, 58982, 74313, 93628, 117964,
                                     148626, 187257, 235929, 297252, 374514, 471859, 594505, 749029, 943718, 1189010, 1498059, 1887436 };
int x264o_lambda(int qp) { return lambda_tab[clampi(qp, 0, 51)]; }
int x264o_lambda2(int qp) { return lambda2_tab[clampi(qp, 0, 51)]; }

static int ref_cost(const actx *a, int r)       /* REF_COST(0, r): lambda * bs_size_te(nref - 1, r) */
{
    return a->nref <= 1 ? 0 : a->lambda * (a->nref == 2 ? 1 : bs_size_ue(r));
}
static int ref_cost_l(const actx *a, int l, int r)       /* REF_COST(l, r) */
{
    return a->nref_l[l] <= 1 ? 0 : a->lambda * (a->nref_l[l] == 2 ? 1 : bs_size_ue(r));
}

static int mbcmp(const actx *a, const pixel *p, int sp, const pixel *q, int sq, int w, int h)
{
    return a->satd ? x264o_satd(p, sp, q, sq, w, h) : x264o_sad(p, sp, q, sq, w, h);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * motion vector prediction (common/mvpred.c) on an 8x8-granular motion cache (the smallest partition here is 8x8) */
static int is_intra_type(int t) { return t >= 0 && t <= 3; }

static nb_t nb8l(const actx *a, int l, int gx, int gy)
{
    nb_t n = { -2, { 0, 0 } };
    const x264o_encoder *e = a->e;
    if (gx < 0 || gy < 2 * e->row0 || gx >= 2 * e->mbw || gy >= 2 * e->mbh) return n;      /* outside the picture or the slice */
    const int i = (gy >> 1) * e->mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
    if (i == a->mi) { if ((l ? a->cur_valid1 : a->cur_valid) >> k & 1) return l ? a->cur8b[k] : a->cur8[k]; return n; }
    if (i > a->mi) return n;
    const x264gpu_mb *m = &e->mbs[i];
    if (is_intra_type(m->type)) { n.ref = -1; return n; }
    /* a block that does not use the list has reference -1 and a zero vector in x264's cache */
    if (l) { n.ref = m->ref1[k]; if (n.ref >= 0) { n.mv[0] = m->mv1[k][0]; n.mv[1] = m->mv1[k][1]; } else n.ref = -1; }
    else { n.ref = m->ref[k]; if (n.ref >= 0) { n.mv[0] = m->mv[k][0]; n.mv[1] = m->mv[k][1]; } else n.ref = -1; }
    return n;
}
static nb_t nb8(const actx *a, int gx, int gy) { return nb8l(a, 0, gx, gy); }

static void median_mv(int mvp[2], const nb_t *A, const nb_t *B, const nb_t *C)
{
    mvp[0] = median3(A->mv[0], B->mv[0], C->mv[0]);
    mvp[1] = median3(A->mv[1], B->mv[1], C->mv[1]);
}

/* x264_mb_predict_mv / _16x16: partition at 8x8 offset (bx8,by8), w8 blocks wide, of reference `ref` */
static void predict_mv_l(const actx *a, int l, int bx8, int by8, int w8, int ref, int mvp[2])
{
    const int gx = 2 * a->mbx + bx8, gy = 2 * a->mby + by8;
    nb_t A = nb8l(a, l, gx - 1, gy), B = nb8l(a, l, gx, gy - 1), C = nb8l(a, l, gx + w8, gy - 1);
    if (C.ref == -2) C = nb8l(a, l, gx - 1, gy - 1);
    if (a->partition == D_16x8) {
        if (by8 == 0) { if (B.ref == ref) { mvp[0] = B.mv[0]; mvp[1] = B.mv[1]; return; } }
        else if (A.ref == ref) { mvp[0] = A.mv[0]; mvp[1] = A.mv[1]; return; }
    } else if (a->partition == D_8x16) {
        if (bx8 == 0) { if (A.ref == ref) { mvp[0] = A.mv[0]; mvp[1] = A.mv[1]; return; } }
        else if (C.ref == ref) { mvp[0] = C.mv[0]; mvp[1] = C.mv[1]; return; }
    }
    const int cnt = (A.ref == ref) + (B.ref == ref) + (C.ref == ref);
    if (cnt > 1) median_mv(mvp, &A, &B, &C);
    else if (cnt == 1) {
        const nb_t *s = A.ref == ref ? &A : B.ref == ref ? &B : &C;
        mvp[0] = s->mv[0]; mvp[1] = s->mv[1];
    } else if (B.ref == -2 && C.ref == -2 && A.ref != -2) { mvp[0] = A.mv[0]; mvp[1] = A.mv[1]; }
    else median_mv(mvp, &A, &B, &C);
}

static void predict_mv(const actx *a, int bx8, int by8, int w8, int ref, int mvp[2]) { predict_mv_l(a, 0, bx8, by8, w8, ref, mvp); }

static void predict_mv_pskip(actx *a, int mv[2])
{
    const int gx = 2 * a->mbx, gy = 2 * a->mby, part = a->partition;
    const nb_t A = nb8(a, gx - 1, gy), B = nb8(a, gx, gy - 1);
    if (A.ref == -2 || B.ref == -2 || (A.ref == 0 && !A.mv[0] && !A.mv[1]) || (B.ref == 0 && !B.mv[0] && !B.mv[1])) { mv[0] = mv[1] = 0; return; }
    a->partition = D_16x16;
    predict_mv(a, 0, 0, 2, 0, mv);
    a->partition = part;
}

/* x264_mb_predict_mv_ref16x16: search candidates of the 16x16 block in reference r */
static int16_t (*mvr_of(const x264o_encoder *e, int l, int r))[2] { return l ? e->mvr1[r] : r == 0 ? e->mv16[e->cur] : e->mvr[r]; }

static int predict_mv_ref16x16_l(const actx *a, int l, int r, int mvc[][2])
{
    const x264o_encoder *e = a->e;
    int16_t (*mvr)[2] = mvr_of(e, l, r);
    int n = 0;
    /* B slices: the direct vector of the last 8x8 block when it points into this reference (h->mb.cache still holds the direct prediction) */
    if (e->slice_type == X264GPU_SLICE_B && a->b_direct_available && a->direct_ref[l][3] == r) { mvc[n][0] = a->direct_mv[l][3][0]; mvc[n][1] = a->direct_mv[l][3][1]; n++; }      /* (h->mb.cache.ref[l][scan8[12]]: the direct prediction's last block) */
    const int16_t *lowres = l ? e->lowres_mv1 : e->lowres_mv;
    if (r == 0 && lowres && lowres[0] != 0x7fff) {          /* h->fenc->lowres_mvs[l][distance - 1] */
        mvc[n][0] = lowres[2 * a->mi] * 2; mvc[n][1] = lowres[2 * a->mi + 1] * 2; n++;
    }
    /* spatial: left, top, top-left, top-right 16x16 results in THIS reference; a missing neighbour reads the zero entry in front of the array */
    const int nbi[4] = { a->type_left >= 0 ? a->mi - 1 : -1, a->type_top >= 0 ? a->mi - e->mbw : -1,
                         a->type_tl >= 0 ? a->mi - e->mbw - 1 : -1, a->type_tr >= 0 ? a->mi - e->mbw + 1 : -1 };
    for (int i = 0; i < 4; i++) {
        mvc[n][0] = nbi[i] >= 0 ? mvr[nbi[i]][0] : 0; mvc[n][1] = nbi[i] >= 0 ? mvr[nbi[i]][1] : 0; n++;
    }
    /* temporal: the co-located macroblock of reference 0 and its right / lower neighbour, scaled by the POC distances */
    const int s0 = ref_slot(e, 0);
    if (e->slot_nref[s0] > 0) {
        const int curpoc = e->poc, refpoc = e->slot_poc[ref_slot_l(e, l, r)];
        const int delta = e->slot_poc[s0] - e->slot_ref0poc[s0];      /* l0's own distance to its reference 0 */
        const int inv = (256 + delta / 2) / delta, scale = (curpoc - refpoc) * inv;
        const int16_t (*l0)[2] = e->mv16[s0];
        const int at[3] = { a->mi, a->mbx < e->mbw - 1 ? a->mi + 1 : -1, a->mby < e->mbh - 1 ? a->mi + e->mbw : -1 };
        for (int i = 0; i < 3; i++)
            if (at[i] >= 0) { mvc[n][0] = (l0[at[i]][0] * scale + 128) >> 8; mvc[n][1] = (l0[at[i]][1] * scale + 128) >> 8; n++; }
    }
    return n;
}
static int predict_mv_ref16x16(const actx *a, int r, int mvc[][2]) { return predict_mv_ref16x16_l(a, 0, r, mvc); }

/* ---------------------------------------------------------------------------------------------------------------------------
 * motion search (encoder/me.c) */
typedef struct {
    const actx *a; me_t *m;
    const pixel *fenc; pixel *planes[4]; const pixel *full;
    const uint16_t *cmx, *cmy;
    int refslot;
    int wt, wdenom, wscale, woffset;       /* m->weight[0]: the explicit luma weight of a P slice's list-0 index (--weightp) */
} sctx;

/* full-pel candidates read the weighted copy of the plane (x264: m->p_fref_w = h->fenc->weighted[ref], x264_weight_scale_plane of the padded
 * plane — pointwise, so weighting the block read from the unweighted plane gives the same samples) */
static int cost_fpel(const sctx *s, int mx, int my)
{
    const x264o_encoder *e = s->a->e;
    if (s->wt) {
        pixel wb[256];
        x264o_mc_weight(wb, 16, s->full + my * e->rs + mx, e->rs, s->m->w, s->m->h, s->wscale, s->wdenom, s->woffset);
        return x264o_sad(s->fenc, e->fs, wb, 16, s->m->w, s->m->h) + s->cmx[mx * 4] + s->cmy[my * 4];
    }
    return x264o_sad(s->fenc, e->fs, s->full + my * e->rs + mx, e->rs, s->m->w, s->m->h) + s->cmx[mx * 4] + s->cmy[my * 4];
}
/* mc.get_ref: interpolate, then weight (common/mc.c get_ref: mc_weight after pixel_avg / on the copied block) */
static void get_ref(const sctx *s, pixel *dst, int mx, int my)
{
    const actx *a = s->a;
    x264o_mc_luma(dst, 16, s->planes, a->e->rs, a->mbx * 16 + s->m->ox, a->mby * 16 + s->m->oy, mx, my, s->m->w, s->m->h);
    if (s->wt) x264o_mc_weight(dst, 16, dst, 16, s->m->w, s->m->h, s->wscale, s->wdenom, s->woffset);
}
static int cost_qpel_sad(const sctx *s, int mx, int my)       /* COST_MV_HPEL / COST_MV_SAD */
{
    pixel pred[256];
    get_ref(s, pred, mx, my);
    return x264o_sad(s->fenc, s->a->e->fs, pred, 16, s->m->w, s->m->h) + s->cmx[mx] + s->cmy[my];
}
static int chroma_satd(const sctx *s, int mx, int my)
{
    const actx *a = s->a; const x264o_encoder *e = a->e; const me_t *m = s->m;
    pixel pu[64], pv[64], fu[64], fv[64];
    const pixel *fuv = e->fenc_uv + (size_t)(a->mby * 8 + m->oy / 2) * e->fs + a->mbx * 16 + m->ox;
    for (int y = 0; y < m->h / 2; y++)
        for (int x = 0; x < m->w / 2; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    x264o_mc_chroma(pu, pv, 8, chroma_plane(e, s->refslot), e->rs, a->mbx * 8 + m->ox / 2, a->mby * 8 + m->oy / 2, mx, my, m->w / 2, m->h / 2);
    if (e->slice_type == X264GPU_SLICE_P && m->list == 0) {          /* m->weight[1] / [2] */
        if (e->wc0[m->ref].on[0]) x264o_mc_weight(pu, 8, pu, 8, m->w / 2, m->h / 2, e->wc0[m->ref].scale[0], e->wc0[m->ref].denom, e->wc0[m->ref].offset[0]);
        if (e->wc0[m->ref].on[1]) x264o_mc_weight(pv, 8, pv, 8, m->w / 2, m->h / 2, e->wc0[m->ref].scale[1], e->wc0[m->ref].denom, e->wc0[m->ref].offset[1]);
    }
    return mbcmp(a, fu, 8, pu, 8, m->w / 2, m->h / 2) + mbcmp(a, fv, 8, pv, 8, m->w / 2, m->h / 2);
}

static const int8_t hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
static const int8_t square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
static const int8_t mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const uint8_t subpel_iterations[12][4] = { { 0, 0, 0, 0 }, { 1, 1, 0, 0 }, { 0, 1, 1, 0 }, { 0, 2, 1, 0 }, { 0, 2, 1, 1 }, { 0, 2, 1, 2 },
                                                  { 0, 0, 2, 2 }, { 0, 0, 2, 2 }, { 0, 0, 4, 10 }, { 0, 0, 4, 10 }, { 0, 0, 4, 10 }, { 0, 0, 4, 10 } };

static void refine_subpel(const sctx *s, int hpel_iters, int qpel_iters, int *p_halfpel_thresh, int b_refine_qpel)
{
    const actx *a = s->a; me_t *m = s->m;
    const int b_chroma_me = a->chroma_me;               /* blocks here are 8x8 or larger */
    int bmx = m->mv[0], bmy = m->mv[1], bcost = m->cost, odir = -1, bdir;
    pixel pred[256];
    const x264o_encoder *e = a->e;
    if (hpel_iters) {
        if (a->subme < 3) {     /* try the sub-pel component of the predicted vector */
            const int mx = clampi(m->mvp[0], a->smin[0] + 2, a->smax[0] - 2), my = clampi(m->mvp[1], a->smin[1] + 2, a->smax[1] - 2);
            if ((mx - bmx) | (my - bmy)) { const int c = cost_qpel_sad(s, mx, my); if (c < bcost) { bcost = c; bmx = mx; bmy = my; } }
        }
        static const int8_t dia[4][2] = { { 0, -2 }, { 0, 2 }, { -2, 0 }, { 2, 0 } };
        for (int i = hpel_iters; i > 0; i--) {
            const int omx = bmx, omy = bmy;
            int best = -1;
            for (int k = 0; k < 4; k++) {
                const int c = cost_qpel_sad(s, omx + dia[k][0], omy + dia[k][1]);
                if (c < bcost) { bcost = c; best = k; }
            }
            if (best < 0) break;
            bmx = omx + dia[best][0]; bmy = omy + dia[best][1];
        }
    }
#define COST_MV_SATD(mx, my, dir) \
    if (b_refine_qpel || ((dir) ^ 1) != odir) { \
        get_ref(s, pred, mx, my); \
        int cost_ = mbcmp(a, s->fenc, e->fs, pred, 16, m->w, m->h) + s->cmx[mx] + s->cmy[my]; \
        if (b_chroma_me && cost_ < bcost) cost_ += chroma_satd(s, mx, my); \
        if (cost_ < bcost) { bcost = cost_; bmx = mx; bmy = my; bdir = dir; } \
    }
    if (!b_refine_qpel && (a->satd || b_chroma_me)) {
        bcost = COST_MAX;
        COST_MV_SATD(bmx, bmy, -1);
    }
    /* early termination when examining several reference frames */
    if (p_halfpel_thresh) {
        if ((bcost * 7) >> 3 > *p_halfpel_thresh) { m->cost = bcost; m->mv[0] = bmx; m->mv[1] = bmy; return; }
        else if (bcost < *p_halfpel_thresh) *p_halfpel_thresh = bcost;
    }
    if (a->subme != 1) {
        bdir = -1;
        for (int i = qpel_iters; i > 0; i--) {
            if (bmy <= a->smin[1] || bmy >= a->smax[1] || bmx <= a->smin[0] || bmx >= a->smax[0]) break;
            odir = bdir;
            const int omx = bmx, omy = bmy;
            COST_MV_SATD(omx, omy - 1, 0);
            COST_MV_SATD(omx, omy + 1, 1);
            COST_MV_SATD(omx - 1, omy, 2);
            COST_MV_SATD(omx + 1, omy, 3);
            if (bmx == omx && bmy == omy) break;
        }
    } else if (bmy > a->smin[1] && bmy < a->smax[1] && bmx > a->smin[0] && bmx < a->smax[0]) {
        /* subme 1: one quarter-pel step on SAD */
        static const int8_t qd[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        const int omx = bmx, omy = bmy;
        for (int k = 0; k < 4; k++) {
            const int c = cost_qpel_sad(s, omx + qd[k][0], omy + qd[k][1]);
            if (c < bcost) { bcost = c; bmx = omx + qd[k][0]; bmy = omy + qd[k][1]; }
        }
    }
#undef COST_MV_SATD
    m->cost = bcost; m->mv[0] = bmx; m->mv[1] = bmy;
    m->cost_mv = s->cmx[bmx] + s->cmy[bmy];
}

static void sctx_init(sctx *s, const actx *a, me_t *m)
{
    const x264o_encoder *e = a->e;
    s->a = a; s->m = m;
    s->fenc = e->fenc_y + (size_t)(a->mby * 16 + m->oy) * e->fs + a->mbx * 16 + m->ox;
    s->refslot = ref_slot_l(e, m->list, m->ref);
    for (int k = 0; k < 4; k++) s->planes[k] = luma_plane(e, s->refslot, k);
    s->full = s->planes[0] + (size_t)(a->mby * 16 + m->oy) * e->rs + a->mbx * 16 + m->ox;
    s->cmx = a->cost_mv - m->mvp[0]; s->cmy = a->cost_mv - m->mvp[1];
    s->wt = e->slice_type == X264GPU_SLICE_P && m->list == 0 && e->wl0[m->ref].on;
    s->wdenom = e->wl0[m->ref].denom; s->wscale = e->wl0[m->ref].scale; s->woffset = e->wl0[m->ref].offset;
}

/* x264_me_search_ref: m->{w,h,ox,oy,ref,mvp} set by the caller */
static void me_search_ref(const actx *a, me_t *m, int (*mvc)[2], int i_mvc, int *p_halfpel_thresh)
{
    const x264o_encoder *e = a->e;
    sctx S, *s = &S;
    sctx_init(s, a, m);
    int i_me_range = e->cfg.me_range;
    int bmx, bmy, bcost = COST_MAX, bpred_cost = COST_MAX, bpred_mx = 0, bpred_my = 0, pmx, pmy, pmv_nonzero, pmv_is_bpred_fullpel = 0;
    const int *fmin = a->fmin, *fmax = a->fmax;
    int cand[16][2];
#define COST_MV(mx, my) do { const int c_ = cost_fpel(s, mx, my); if (c_ < bcost) { bcost = c_; bmx = (mx); bmy = (my); } } while (0)
#define CHECK_MVRANGE(mx, my) ((mx) >= fmin[0] && (mx) <= fmax[0] && (my) >= fmin[1] && (my) <= fmax[1])
    if (a->subme >= 3) {
        /* the predictor and the candidates at sub-pel precision (SAD of the interpolated block) */
        bpred_mx = clampi(m->mvp[0], fmin[0] * 4, fmax[0] * 4); bpred_my = clampi(m->mvp[1], fmin[1] * 4, fmax[1] * 4);
        const int pmvx = bpred_mx, pmvy = bpred_my;
        pmv_nonzero = (pmvx | pmvy) != 0;
        pmx = (pmvx + 2) >> 2; pmy = (pmvy + 2) >> 2;
        bpred_cost = cost_qpel_sad(s, bpred_mx, bpred_my);
        const int pmv_cost = bpred_cost;
        if (i_mvc > 0) {
            /* x264_predictor_clip: drop candidates equal to zero or to pmv (before clipping), clip the rest */
            int n = 0;
            for (int i = 0; i < i_mvc; i++) {
                const int mx = mvc[i][0], my = mvc[i][1];
                if (!(mx | my) || (mx == pmvx && my == pmvy)) continue;
                cand[n][0] = clampi(mx, fmin[0] * 4, fmax[0] * 4); cand[n][1] = clampi(my, fmin[1] * 4, fmax[1] * 4); n++;
            }
            for (int i = 0; i < n; i++) {            /* first strictly better candidate in order wins */
                const int c = cost_qpel_sad(s, cand[i][0], cand[i][1]);
                if (c < bpred_cost) { bpred_cost = c; bpred_mx = cand[i][0]; bpred_my = cand[i][1]; }
            }
        }
        /* back to full-pel: the search starts at the rounded best predictor */
        bmx = (bpred_mx + 2) >> 2; bmy = (bpred_my + 2) >> 2;
        if ((bpred_mx & 3) | (bpred_my & 3)) { bcost = COST_MAX; COST_MV(bmx, bmy); }
        else bcost = bpred_cost;
        if (pmv_nonzero) { if (bmx | bmy) COST_MV(0, 0); }
        else if (pmv_cost < bcost) { bcost = pmv_cost; bmx = 0; bmy = 0; }
    } else {
        /* full-pel predictor, costed without its vector bits */
        bmx = pmx = clampi((m->mvp[0] + 2) >> 2, fmin[0], fmax[0]); bmy = pmy = clampi((m->mvp[1] + 2) >> 2, fmin[1], fmax[1]);
        pmv_nonzero = (pmx | pmy) != 0;
        bcost = cost_fpel(s, bmx, bmy) - s->cmx[bmx * 4] - s->cmy[bmy * 4];
        if (i_mvc > 0) {
            /* x264_predictor_roundclip: round to full-pel, clip, drop zero and pmv */
            int n = 0;
            for (int i = 0; i < i_mvc; i++) {
                const int mx = clampi((mvc[i][0] + 2) >> 2, fmin[0], fmax[0]), my = clampi((mvc[i][1] + 2) >> 2, fmin[1], fmax[1]);
                if (!(mx | my) || (mx == pmx && my == pmy)) continue;
                cand[n][0] = mx; cand[n][1] = my; n++;
            }
            for (int i = 0; i < n; i++) {
                const int c = cost_fpel(s, cand[i][0], cand[i][1]);
                if (c < bcost) { bcost = c; bmx = cand[i][0]; bmy = cand[i][1]; }
            }
        }
        if (pmv_nonzero) COST_MV(0, 0);
        pmv_is_bpred_fullpel = 1;
    }

    switch (e->cfg.me_method) {
    case 0: {   /* X264_ME_DIA: radius-1 diamond, (0,-1) (0,1) (-1,0) (1,0); the centre wins ties */
        static const int8_t dia1[4][2] = { { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 } };
        int i = i_me_range;
        do {
            int best = -1;
            const int omx = bmx, omy = bmy;
            for (int k = 0; k < 4; k++) { const int c = cost_fpel(s, omx + dia1[k][0], omy + dia1[k][1]); if (c < bcost) { bcost = c; best = k; } }
            if (best < 0) break;
            bmx = omx + dia1[best][0]; bmy = omy + dia1[best][1];
        } while (--i && CHECK_MVRANGE(bmx, bmy));
        break;
    }
    case 3: {   /* X264_ME_ESA: every position of the clipped +-merange rectangle (width rounded up to 4), raster order */
        const int min_x = bmx - i_me_range > fmin[0] ? bmx - i_me_range : fmin[0], min_y = bmy - i_me_range > fmin[1] ? bmy - i_me_range : fmin[1];
        const int max_x = bmx + i_me_range < fmax[0] ? bmx + i_me_range : fmax[0], max_y = bmy + i_me_range < fmax[1] ? bmy + i_me_range : fmax[1];
        const int width = (max_x - min_x + 3) & ~3;
        for (int my = min_y; my <= max_y; my++)
            for (int mx = min_x; mx < min_x + width && mx <= fmax[0]; mx++) COST_MV(mx, my);      /* x264 also costs the <= 3 positions of the
                                                                                                    * rounded-up width beyond the limit; skipped here (they read past the picture padding) */
        break;
    }
    case 2: {   /* X264_ME_UMH */
        static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
        static const int8_t hex4[16][2] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                            { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };
        const int shift = (m->w == 16 ? 0 : 1) + (m->h == 16 ? 0 : 1);          /* pixel_size_shift */
        int omx, omy, cross_start = 1, done = 0;
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
                const int r1 = (i_me_range >> 1) | 1;
                CROSS(3, r1, r1);
                COST_MV_X4(-1, -2, 1, -2, -2, -1, 2, -1);
                COST_MV_X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2) done = 1;
                cross_start = r1 + 2;
            }
        }
        if (done) goto fullpel_done;
        if (i_mvc) {        /* adaptive search range: agreement of the predictors x SAD level */
            int mvd, denom = 1;
            if (i_mvc == 1) {
                if (m->w == 16 && m->h == 16) mvd = 25;
                else mvd = abs(m->mvp[0] - mvc[0][0]) + abs(m->mvp[1] - mvc[0][1]);
            } else {
                denom = i_mvc - 1; mvd = 0;
                if (!(m->w == 16 && m->h == 16)) { mvd = abs(m->mvp[0] - mvc[0][0]) + abs(m->mvp[1] - mvc[0][1]); denom++; }
                for (int i = 0; i < i_mvc - 1; i++) mvd += abs(mvc[i][0] - mvc[i + 1][0]) + abs(mvc[i][1] - mvc[i + 1][1]);      /* x264_predictor_difference */
            }
            const int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            const int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            i_me_range = i_me_range * range_mul[mvd_ctx][sad_ctx] >> 2;
        }
        /* x264 keeps the cross centred where the small diamonds left it */
        CROSS(cross_start, i_me_range, i_me_range >> 1);
        COST_MV_X4(-2, -2, -2, 2, 2, -2, 2, 2);
        omx = bmx; omy = bmy;
        {
            int i = 1;
            do {
                for (int j = 0; j < 16; j++) {
                    const int mx = omx + hex4[j][0] * i, my = omy + hex4[j][1] * i;
                    if (CHECK_MVRANGE(mx, my)) COST_MV(mx, my);
                }
            } while (++i <= i_me_range >> 2);
        }
        if (!CHECK_MVRANGE(bmx, bmy)) goto fullpel_done;
#undef SAD_THRESH
#undef CROSS
#undef DIA1_ITER
#undef COST_MV_X4
    }   /* fall through: me_hex2 */
    /* FALLTHROUGH */
    default: {  /* X264_ME_HEX: hexagon (radius 2), then 3x3 square refine; first-best wins ties, the centre wins over all */
        int key = bcost << 3;
        for (int k = 1; k <= 6; k++) {
            const int c = (cost_fpel(s, bmx + hex2[k][0], bmy + hex2[k][1]) << 3) + k + 1;
            if (c < key) key = c;
        }
        if (key & 7) {
            int dir = (key & 7) - 2;
            bmx += hex2[dir + 1][0]; bmy += hex2[dir + 1][1];
            for (int i = (i_me_range >> 1) - 1; i > 0 && CHECK_MVRANGE(bmx, bmy); i--) {
                key &= ~7;
                for (int k = 0; k < 3; k++) {
                    const int c = (cost_fpel(s, bmx + hex2[dir + k][0], bmy + hex2[dir + k][1]) << 3) + k + 1;
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
            const int c = cost_fpel(s, bmx + square1[k][0], bmy + square1[k][1]);
            if (c < bcost) { bcost = c; bdir = k; }
        }
        bmx += square1[bdir][0]; bmy += square1[bdir][1];
        break;
    }
    }
fullpel_done:
#undef COST_MV
#undef CHECK_MVRANGE
    /* -> quarter-pel vector */
    if (a->subme < 3) {
        m->cost_mv = s->cmx[bmx * 4] + s->cmy[bmy * 4];
        m->cost = bcost;
        if (pmv_is_bpred_fullpel && bmx == pmx && bmy == pmy) m->cost += m->cost_mv;        /* the real cost */
        m->mv[0] = bmx * 4; m->mv[1] = bmy * 4;
    } else {
        if (bpred_cost < bcost) { m->mv[0] = bpred_mx; m->mv[1] = bpred_my; m->cost = bpred_cost; }
        else { m->mv[0] = bmx * 4; m->mv[1] = bmy * 4; m->cost = bcost; }
    }
    m->cost_mv = s->cmx[m->mv[0]] + s->cmy[m->mv[1]];
    if (a->subme >= 2) refine_subpel(s, subpel_iterations[a->subme][2], subpel_iterations[a->subme][3], p_halfpel_thresh, 0);
}

/* x264_me_refine_qpel: the extra sub-pel steps of the winning partition below subme 6 */
/* x264_me_refine_qpel_refdupe: the blind duplicate of reference 0 (--weightp 2) is never searched — its vector starts at reference 0's result
 * (set by the caller) and takes at most two quarter-pel iterations.  m->cost is whatever the previous search left (it only matters below subme 2,
 * where the starting cost is not recomputed — as in x264, whose x264_me_t is reused across the reference loop) */
static void me_refine_qpel_refdupe(const actx *a, me_t *m, int *p_halfpel_thresh)
{
    sctx S;
    sctx_init(&S, a, m);
    const int q = subpel_iterations[a->subme][3];
    refine_subpel(&S, 0, q < 2 ? q : 2, p_halfpel_thresh, 0);
}

static void me_refine_qpel(const actx *a, me_t *m)
{
    sctx S;
    sctx_init(&S, a, m);
    m->cost -= m->ref_cost;                   /* blocks of 8x8 and larger */
    refine_subpel(&S, subpel_iterations[a->subme][0], subpel_iterations[a->subme][1], NULL, 1);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * residual coding (encoder/macroblock.c) */
static void scan4(int16_t *dst, const dctcoef *src) { for (int k = 0; k < 16; k++) dst[k] = src[x264o_zigzag4[k]]; }

/* The quantiser calls of x264_macroblock_encode: the dead-zone quantiser, or — in the final encode of a macroblock of a trellis session
 * (e->b_trellis; x264 h->mb.b_trellis under --trellis 1) — the trellis search of trellis.cpp on the slice's live context variables.
 * cat = CABAC block category, qp = the block's quantiser (chroma: the chroma one) */
int x264o_quant_trellis_cabac(dctcoef *dct, const uint16_t *mf, int qp, int cat, int intra, const uint8_t *state);
/* e->b_trellis: the sites of the final encode that use it, as a mask (cfg.trellis; x264's --trellis 1 = all of them = 63): 1 inter luma 4x4,
 * 2 inter luma 8x8, 4 chroma, 8 Intra_16x16, 16 Intra_4x4, 32 Intra_8x8 — the device was brought up site by site against these masks */
enum { TR_P4 = 1, TR_P8 = 2, TR_C = 4, TR_I16 = 8, TR_I4 = 16, TR_I8 = 32 };
static int quant_4x4(x264o_encoder *e, dctcoef *d, const uint16_t *mf, const uint16_t *bias, int qp, int cat, int intra)
{
    const int site = cat == 4 ? TR_C : cat == 1 ? TR_I16 : intra ? TR_I4 : TR_P4;
    return (e->b_trellis & site) ? x264o_quant_trellis_cabac(d, mf, qp, cat, intra, e->cabac_state) : x264o_quant_4x4(d, mf, bias);
}
static int quant_8x8(x264o_encoder *e, dctcoef *d, const uint16_t *mf, const uint16_t *bias, int qp, int intra)
{
    return (e->b_trellis & (intra ? TR_I8 : TR_P8)) ? x264o_quant_trellis_cabac(d, mf, qp, 5, intra, e->cabac_state) : x264o_quant_8x8(d, mf, bias);
}
static int quant_4x4_dc(x264o_encoder *e, dctcoef *d, const uint16_t *mf, const uint16_t *bias, int qp)
{
    return (e->b_trellis & TR_I16) ? x264o_quant_trellis_cabac(d, mf, qp, 0, 1, e->cabac_state) : x264o_quant_4x4_dc(d, mf[0] >> 1, bias[0] << 1);
}
static int quant_2x2_dc(x264o_encoder *e, dctcoef *d, const uint16_t *mf, const uint16_t *bias, int qp, int intra)
{
    return (e->b_trellis & TR_C) ? x264o_quant_trellis_cabac(d, mf, qp, 3, intra, e->cabac_state) : x264o_quant_2x2_dc(d, mf[0] >> 1, bias[0] << 1);
}

/* inter luma, 4x4 transform: prediction in rec; returns through mb: nnz, cbp_luma; levels in lv */
static void encode_luma_inter(x264o_encoder *e, const pixel *fenc, pixel *rec, int qp, x264gpu_mb *mb, int16_t *lv)
{
    dctcoef d[16][16];
    int nz[16], score8[4] = { 0, 0, 0, 0 };
    const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4PY][qp], *bias = e->qt.quant4_bias[X264O_CQM_4PY][qp];
    for (int b = 0; b < 16; b++) {
        x264o_sub4x4_dct(d[b], fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4, e->fs, rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs);
        nz[b] = quant_4x4(e, d[b], mf, bias, qp, 2, 0);
        scan4(lv + b * 16, d[b]);
        if (nz[b] && e->cfg.dct_decimate) score8[b >> 2] += x264o_decimate_score(lv + b * 16, 16);
    }
    int mbscore = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        int any = nz[i8 * 4] | nz[i8 * 4 + 1] | nz[i8 * 4 + 2] | nz[i8 * 4 + 3];
        if (any) mbscore += e->cfg.dct_decimate ? score8[i8] : 6;      /* every coded 8x8 counts towards the macroblock score, kept or not */
        if (any && e->cfg.dct_decimate && score8[i8] < 4) any = 0;
        if (!any) for (int k = 0; k < 4; k++) nz[i8 * 4 + k] = 0;
    }
    if (mbscore < 6) for (int b = 0; b < 16; b++) nz[b] = 0;
    for (int b = 0; b < 16; b++) {
        if (!nz[b]) { memset(lv + b * 16, 0, 32); continue; }
        x264o_dequant_4x4(d[b], e->qt.dequant4_mf, qp);
        x264o_add4x4_idct(rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs, d[b]);
        mb->nnz |= 1u << b;
        mb->cbp_luma |= 1 << (b >> 2);
    }
}

/* inter luma, 8x8 transform; levels leave in the CAVLC-interleaved 4x4 form (zigzag_interleave_8x8_cavlc) */
static void encode_luma_inter8(x264o_encoder *e, const pixel *fenc, pixel *rec, int qp, x264gpu_mb *mb, int16_t *lv)
{
    dctcoef d[4][64];
    int16_t scan[4][64];
    int keep[4], mbscore = 0;
    const uint16_t *mf = e->qt.quant8_mf[X264O_CQM_8PY][qp], *bias = e->qt.quant8_bias[X264O_CQM_8PY][qp];
    /* x264_macroblock_encode: "b_decimate &= !h->mb.b_trellis || !h->param.b_cabac; 8x8 trellis is inherently optimal decimation for CABAC" */
    const int b_decimate = e->cfg.dct_decimate && !(e->b_trellis & TR_P8);
    for (int i8 = 0; i8 < 4; i8++) {
        x264o_sub8x8_dct8(d[i8], fenc + (i8 >> 1) * 8 * e->fs + (i8 & 1) * 8, e->fs, rec + (i8 >> 1) * 8 * e->rs + (i8 & 1) * 8, e->rs);
        keep[i8] = quant_8x8(e, d[i8], mf, bias, qp, 0);
        for (int k = 0; k < 64; k++) scan[i8][k] = d[i8][x264o_zigzag8[k]];
        if (keep[i8] && b_decimate) {
            const int sc = x264o_decimate_score(scan[i8], 64);
            mbscore += sc;
            if (sc < 4) keep[i8] = 0;
        }
    }
    if (b_decimate && mbscore < 6) keep[0] = keep[1] = keep[2] = keep[3] = 0;
    for (int i8 = 0; i8 < 4; i8++) {
        if (!keep[i8]) continue;
        for (int k = 0; k < 64; k++) {
            const int16_t v = scan[i8][k];
            lv[(i8 * 4 + (k & 3)) * 16 + (k >> 2)] = v;
            if (v) mb->nnz |= 1u << (i8 * 4 + (k & 3));
        }
        x264o_dequant_8x8(d[i8], e->qt.dequant8_mf, qp);
        x264o_add8x8_idct8(rec + (i8 >> 1) * 8 * e->rs + (i8 & 1) * 8, e->rs, d[i8]);
        mb->cbp_luma |= 1 << i8;
    }
}

/* x264_mb_encode_chroma: prediction already in the NV12 reconstruction (rec_uv points at U of the 8x8) */
static void encode_chroma(x264o_encoder *e, const pixel *fenc_uv, pixel *rec_uv, int qpc, int inter, x264gpu_mb *mb, int16_t *lv)
{
    const int list = inter ? X264O_CQM_4PC : X264O_CQM_4IC, b_decimate = inter && e->cfg.dct_decimate;
    const uint16_t *mf = e->qt.quant4_mf[list][qpc], *bias = e->qt.quant4_bias[list][qpc];
    const int dmf = e->qt.dequant4_mf[qpc % 6][0] << (qpc / 6);
    pixel f[2][64], p[2][64];
    int any_ac = 0, any_dc = 0;
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) { f[c][y * 8 + x] = fenc_uv[y * e->fs + 2 * x + c]; p[c][y * 8 + x] = rec_uv[y * e->rs + 2 * x + c]; }
    memset(lv + X264GPU_LV_CHROMA_DC, 0, (8 + 128) * sizeof(int16_t));
    /* early termination on the variance of the chroma residual (not at low quantisers): DC only, or nothing */
    if (b_decimate && qpc >= 18) {
        const int thresh = (x264o_lambda2(qpc) + 32) >> 6;
        int ssd[2], score = 0;
        for (int c = 0; c < 2; c++) {
            int sum = 0, sqr = 0;
            for (int i = 0; i < 64; i++) { const int dd = f[c][i] - p[c][i]; sum += dd; sqr += dd * dd; }
            ssd[c] = sqr;
            score += sqr - (int)(((int64_t)sum * sum) >> 6);
        }
        if (score < thresh * 4) {
            for (int c = 0; c < 2; c++) {
                if (ssd[c] <= thresh) continue;
                dctcoef dc[4];
                for (int i = 0; i < 4; i++) {           /* sub8x8_dct_dc: sums of the four 4x4 residual blocks, then the 2x2 Hadamard */
                    int sm = 0;
                    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { const int o = ((i >> 1) * 4 + y) * 8 + (i & 1) * 4 + x; sm += f[c][o] - p[c][o]; }
                    dc[i] = (dctcoef)sm;
                }
                x264o_dct2x2dc(dc);
                if (!quant_2x2_dc(e, dc, mf, bias, qpc, !inter)) continue;
                if (!x264o_optimize_chroma_2x2_dc(dc, dmf)) continue;
                for (int i = 0; i < 4; i++) lv[X264GPU_LV_CHROMA_DC + c * 4 + i] = dc[i];
                dctcoef dq[4];
                x264o_dequant_2x2_dc(dq, dc, e->qt.dequant4_mf, qpc);
                for (int i = 0; i < 4; i++) x264o_add4x4_idct_dc(p[c] + (i >> 1) * 32 + (i & 1) * 4, 8, dq[i]);
                mb->nnz |= 1u << (25 + c); any_dc = 1;
            }
            goto store;
        }
    }
    for (int c = 0; c < 2; c++) {
        dctcoef d[4][16], dc[4];
        int nz[4], score = 0, nzac = 0;
        for (int i = 0; i < 4; i++) {
            const int o = (i >> 1) * 32 + (i & 1) * 4;
            x264o_sub4x4_dct(d[i], f[c] + o, 8, p[c] + o, 8);
            dc[i] = d[i][0]; d[i][0] = 0;
            nz[i] = quant_4x4(e, d[i], mf, bias, qpc, 4, !inter);
            int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
            scan4(l, d[i]);
            if (nz[i]) { nzac = 1; if (b_decimate) score += x264o_decimate_score(l + 1, 15); }
        }
        if (nzac && b_decimate && score < 7) nzac = 0;
        x264o_dct2x2dc(dc);
        int nzdc = quant_2x2_dc(e, dc, mf, bias, qpc, !inter);
        /* DC-only planes: x264_mb_optimize_chroma_dc trims the DC levels that do not change the reconstruction */
        if (nzdc && !nzac && !x264o_optimize_chroma_2x2_dc(dc, dmf)) { nzdc = 0; dc[0] = dc[1] = dc[2] = dc[3] = 0; }
        for (int i = 0; i < 4; i++) lv[X264GPU_LV_CHROMA_DC + c * 4 + i] = dc[i];
        dctcoef dq[4] = { 0, 0, 0, 0 };
        if (nzdc) { x264o_dequant_2x2_dc(dq, dc, e->qt.dequant4_mf, qpc); mb->nnz |= 1u << (25 + c); any_dc = 1; }
        for (int i = 0; i < 4; i++) {
            int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
            if (!nzac || !nz[i]) { memset(l, 0, 32); memset(d[i], 0, sizeof(d[i])); }
            else { x264o_dequant_4x4(d[i], e->qt.dequant4_mf, qpc); mb->nnz |= 1u << (16 + c * 4 + i); any_ac = 1; }
            d[i][0] = dq[i];
            x264o_add4x4_idct(p[c] + (i >> 1) * 32 + (i & 1) * 4, 8, d[i]);
        }
    }
store:
    for (int c = 0; c < 2; c++)
        for (int y = 0; y < 8; y++)
            for (int x = 0; x < 8; x++) rec_uv[y * e->rs + 2 * x + c] = p[c][y * 8 + x];
    mb->cbp_chroma = any_ac ? 2 : any_dc ? 1 : 0;
}

/* x264_mb_encode_i4x4 / _i8x8: prediction already in rec; returns non-zero when coefficients were coded */
static int encode_i4x4(x264o_encoder *e, const pixel *f, pixel *r, int qp, int16_t *l)
{
    dctcoef d[16];
    x264o_sub4x4_dct(d, f, e->fs, r, e->rs);
    if (!quant_4x4(e, d, e->qt.quant4_mf[X264O_CQM_4IY][qp], e->qt.quant4_bias[X264O_CQM_4IY][qp], qp, 2, 1)) { memset(l, 0, 32); return 0; }
    scan4(l, d);
    x264o_dequant_4x4(d, e->qt.dequant4_mf, qp);
    x264o_add4x4_idct(r, e->rs, d);
    return 1;
}
static int encode_i8x8(x264o_encoder *e, const pixel *f, pixel *r, int qp, int i8, int16_t *lv256, uint32_t *nnz)
{
    dctcoef d[64];
    x264o_sub8x8_dct8(d, f, e->fs, r, e->rs);
    if (!quant_8x8(e, d, e->qt.quant8_mf[X264O_CQM_8IY][qp], e->qt.quant8_bias[X264O_CQM_8IY][qp], qp, 1)) return 0;
    for (int k = 0; k < 64; k++) {
        const int16_t v = d[x264o_zigzag8[k]];
        lv256[(i8 * 4 + (k & 3)) * 16 + (k >> 2)] = v;
        if (v) *nnz |= 1u << (i8 * 4 + (k & 3));
    }
    x264o_dequant_8x8(d, e->qt.dequant8_mf, qp);
    x264o_add8x8_idct8(r, e->rs, d);
    return 1;
}

/* x264_mb_encode_i16x16: prediction of `mode` is written to rec first */
static void encode_i16x16(x264o_encoder *e, const pixel *fenc, pixel *rec, int qp, int mode, x264gpu_mb *mb, int16_t *lv)
{
    pixel pred[256];
    x264o_predict_16x16(pred, 16, rec, e->rs, mode);
    for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, pred + y * 16, 16);
    const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4IY][qp], *bias = e->qt.quant4_bias[X264O_CQM_4IY][qp];
    const int b_decimate = e->slice_type != X264GPU_SLICE_I && e->cfg.dct_decimate;        /* h->mb.b_dct_decimate */
    dctcoef d[16][16], dc[16];
    int nz[16], any_ac = 0, score = b_decimate ? 0 : 9;
    for (int b = 0; b < 16; b++) {
        x264o_sub4x4_dct(d[b], fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4, e->fs, rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs);
        dc[blk_y[b] * 4 + blk_x[b]] = d[b][0]; d[b][0] = 0;
        nz[b] = quant_4x4(e, d[b], mf, bias, qp, 1, 1);
        scan4(lv + b * 16, d[b]);
        if (nz[b]) { any_ac = 1; x264o_dequant_4x4(d[b], e->qt.dequant4_mf, qp); if (score < 6) score += x264o_decimate_score(lv + b * 16 + 1, 15); }
    }
    /* the 16 coded-block flags of an Intra16x16 macroblock are costly: decimate the AC part as a whole */
    if (score < 6) any_ac = 0;
    for (int b = 0; b < 16; b++) {
        if (any_ac && nz[b]) mb->nnz |= 1u << b;
        else { memset(lv + b * 16, 0, 32); memset(d[b], 0, sizeof(d[b])); }
    }
    mb->cbp_luma = any_ac ? 15 : 0;
    x264o_dct4x4dc(dc);
    const int nzdc = quant_4x4_dc(e, dc, mf, bias, qp);
    scan4(lv + X264GPU_LV_LUMA_DC, dc);
    if (nzdc) { mb->nnz |= 1u << 24; x264o_idct4x4dc(dc); x264o_dequant_4x4_dc(dc, e->qt.dequant4_mf, qp); }
    for (int b = 0; b < 16; b++) {
        d[b][0] = nzdc ? dc[blk_y[b] * 4 + blk_x[b]] : 0;
        x264o_add4x4_idct(rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4, e->rs, d[b]);
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * P_SKIP probe (x264_macroblock_probe_pskip): would the macroblock code to nothing at the skip vector? */
static void mc_mb(x264o_encoder *e, int mbx, int mby, int bx, int by, int w, int h, int refslot, int mvx, int mvy, pixel *dy, int sy, pixel *du, pixel *dv, int sc)
{
    pixel *planes[4] = { luma_plane(e, refslot, 0), luma_plane(e, refslot, 1), luma_plane(e, refslot, 2), luma_plane(e, refslot, 3) };
    /* x264 mb_mc_0xywh / _1xywh / _01xywh: the vector is clipped to the macroblock's mv_min / mv_max (24 pixels outside the picture) before the
     * fetch.  Searched vectors are inside anyway; INFERRED ones (spatial direct takes a neighbour's vector as it is, the skip vector) can point
     * farther than the padding reaches — inside the replicated border the clipped vector fetches the same samples */
    mvx = clampi(mvx, 4 * (-16 * mbx - 24), 4 * (16 * (e->mbw - mbx - 1) + 24)); mvy = clampi(mvy, 4 * (-16 * mby - 24), 4 * (16 * (e->mbh - mby - 1) + 24));
    x264o_mc_luma(dy + by * sy + bx, sy, planes, e->rs, mbx * 16 + bx, mby * 16 + by, mvx, mvy, w, h);
    x264o_mc_chroma(du + (by / 2) * sc + bx / 2, dv + (by / 2) * sc + bx / 2, sc, chroma_plane(e, refslot), e->rs, mbx * 8 + bx / 2, mby * 8 + by / 2, mvx, mvy, w / 2, h / 2);
}

/* x264_mb_mc of a B macroblock: per 8x8 block from list 0, list 1, or both averaged with the pair's implicit weight (mb_mc_01xywh) */
static void mc_mb_b(x264o_encoder *e, int mbx, int mby, const x264gpu_mb *mb, pixel *dy, int sy, pixel *du, pixel *dv)
{
    for (int k = 0; k < 4; k++) {
        const int bx = (k & 1) * 8, by = (k >> 1) * 8, r0 = mb->ref[k], r1 = mb->ref1[k];
        if (r0 >= 0 && r1 >= 0) {
            pixel y0[64], y1[64], u0[16], v0[16], u1[16], v1[16];
            mc_mb(e, mbx, mby, bx, by, 8, 8, ref_slot_l(e, 0, r0), mb->mv[k][0], mb->mv[k][1], y0 - by * 8 - bx, 8, u0 - (by / 2) * 4 - bx / 2, v0 - (by / 2) * 4 - bx / 2, 4);
            mc_mb(e, mbx, mby, bx, by, 8, 8, ref_slot_l(e, 1, r1), mb->mv1[k][0], mb->mv1[k][1], y1 - by * 8 - bx, 8, u1 - (by / 2) * 4 - bx / 2, v1 - (by / 2) * 4 - bx / 2, 4);
            const int w = e->bipred_weight[r0][r1];
            x264o_pixel_avg_weight(dy + by * sy + bx, sy, y0, 8, y1, 8, 8, 8, w);
            x264o_pixel_avg_weight(du + (by / 2) * 8 + bx / 2, 8, u0, 4, u1, 4, 4, 4, w);
            x264o_pixel_avg_weight(dv + (by / 2) * 8 + bx / 2, 8, v0, 4, v1, 4, 4, 4, w);
        } else if (r0 >= 0) mc_mb(e, mbx, mby, bx, by, 8, 8, ref_slot_l(e, 0, r0), mb->mv[k][0], mb->mv[k][1], dy, sy, du, dv, 8);
        else mc_mb(e, mbx, mby, bx, by, 8, 8, ref_slot_l(e, 1, r1), mb->mv1[k][0], mb->mv1[k][1], dy, sy, du, dv, 8);
    }
}

/* x264_mb_mc_0xywh of a P slice: list-0 index r with its explicit weights (h->sh.weight[r][0..2]) */
static void mc_mb_p(x264o_encoder *e, int mbx, int mby, int bx, int by, int w, int h, int r, int mvx, int mvy, pixel *dy, int sy, pixel *du, pixel *dv, int sc)
{
    mc_mb(e, mbx, mby, bx, by, w, h, ref_slot(e, r), mvx, mvy, dy, sy, du, dv, sc);
    if (e->wl0[r].on) x264o_mc_weight(dy + by * sy + bx, sy, dy + by * sy + bx, sy, w, h, e->wl0[r].scale, e->wl0[r].denom, e->wl0[r].offset);
    if (e->wc0[r].on[0]) x264o_mc_weight(du + (by / 2) * sc + bx / 2, sc, du + (by / 2) * sc + bx / 2, sc, w / 2, h / 2, e->wc0[r].scale[0], e->wc0[r].denom, e->wc0[r].offset[0]);
    if (e->wc0[r].on[1]) x264o_mc_weight(dv + (by / 2) * sc + bx / 2, sc, dv + (by / 2) * sc + bx / 2, sc, w / 2, h / 2, e->wc0[r].scale[1], e->wc0[r].denom, e->wc0[r].offset[1]);
}

/* x264_macroblock_probe_skip_internal: would the residual of the prediction (py stride 16, pu / pv stride 8) code to nothing? */
static int probe_skip_pred(const actx *a, const pixel *py, const pixel *pu, const pixel *pv)
{
    x264o_encoder *e = a->e;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16;
    const uint16_t *mf = e->qt.quant4_mf[X264O_CQM_4PY][a->qp], *bias = e->qt.quant4_bias[X264O_CQM_4PY][a->qp];
    int score = 0;
    for (int b = 0; b < 16; b++) {
        dctcoef d[16]; int16_t sc[16];
        x264o_sub4x4_dct(d, fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4, e->fs, py + blk_y[b] * 64 + blk_x[b] * 4, 16);
        if (!x264o_quant_4x4(d, mf, bias)) continue;
        scan4(sc, d);
        score += x264o_decimate_score(sc, 16);
        if (score >= 6) return 0;
    }
    /* chroma: cheap SSD test first, then DC, then (rarely) the AC decimation score */
    const int qpc = a->qpc, thresh = (x264o_lambda2(qpc) + 32) >> 6;
    mf = e->qt.quant4_mf[X264O_CQM_4PC][qpc]; bias = e->qt.quant4_bias[X264O_CQM_4PC][qpc];
    const pixel *fuv = e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16;
    for (int c = 0; c < 2; c++) {
        pixel f[64];
        const pixel *p = c ? pv : pu;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) f[y * 8 + x] = fuv[y * e->fs + 2 * x + c];
        const int ssd = x264o_ssd(p, 8, f, 8, 8, 8);
        if (ssd < thresh) continue;
        dctcoef dc[4];
        for (int i = 0; i < 4; i++) {
            int sm = 0;
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { const int o = ((i >> 1) * 4 + y) * 8 + (i & 1) * 4 + x; sm += f[o] - p[o]; }
            dc[i] = (dctcoef)sm;
        }
        x264o_dct2x2dc(dc);
        if (x264o_quant_2x2_dc(dc, mf[0] >> 1, bias[0] << 1)) return 0;
        if (ssd < thresh * 4) continue;
        int cscore = 0;
        for (int i = 0; i < 4; i++) {
            dctcoef d[16]; int16_t sc[16];
            const int o = (i >> 1) * 32 + (i & 1) * 4;
            x264o_sub4x4_dct(d, f + o, 8, p + o, 8);
            d[0] = 0;
            if (!x264o_quant_4x4(d, mf, bias)) continue;
            scan4(sc, d);
            cscore += x264o_decimate_score(sc + 1, 15);
            if (cscore >= 7) return 0;
        }
    }
    return 1;
}

static int probe_pskip(const actx *a)
{
    x264o_encoder *e = a->e;
    const int mvx = clampi(a->pskip_mv[0], a->mv_min[0], a->mv_max[0]), mvy = clampi(a->pskip_mv[1], a->mv_min[1], a->mv_max[1]);
    pixel py[256], pu[64], pv[64];
    mc_mb_p(e, a->mbx, a->mby, 0, 0, 16, 16, 0, mvx, mvy, py, 16, pu, pv, 8);
    return probe_skip_pred(a, py, pu, pv);
}

/* x264_macroblock_probe_bskip: the same test on the direct prediction, which x264_mb_mc left in the reconstruction */
static int probe_bskip(const actx *a)
{
    x264o_encoder *e = a->e;
    const pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    const pixel *ruv = chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    pixel py[256], pu[64], pv[64];
    for (int y = 0; y < 16; y++) memcpy(py + y * 16, rec + (size_t)y * e->rs, 16);
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { pu[y * 8 + x] = ruv[y * e->rs + 2 * x]; pv[y * 8 + x] = ruv[y * e->rs + 2 * x + 1]; }
    return probe_skip_pred(a, py, pu, pv);
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * inter analysis (encoder/analyse.c) */
static void cache_block(actx *a, int bx8, int by8, int w8, int h8, int ref, const int mv[2])
{
    for (int y = by8; y < by8 + h8; y++)
        for (int x = bx8; x < bx8 + w8; x++) {
            nb_t *n = &a->cur8[y * 2 + x];
            n->ref = ref;
            if (mv) { n->mv[0] = mv[0]; n->mv[1] = mv[1]; }
            a->cur_valid |= 1 << (y * 2 + x);
        }
}

static int rd_cost_inter(actx *a, int partition, int t8, x264gpu_mb *mb, int16_t *lv);

/* returns 1 when the macroblock was settled as P_SKIP inside the 16x16 search */
static int analyse_inter_p16x16(actx *a)
{
    x264o_encoder *e = a->e;
    me_t m;
    int mvc[8][2];
    int i_halfpel_thresh = 0x7fffffff;
    int *p_halfpel_thresh = (a->b_early_terminate && a->nref > 1) ? &i_halfpel_thresh : NULL;
    m.w = m.h = 16; m.ox = m.oy = 0; m.list = 0;
    a->me16.cost = 0x7fffffff;
    a->partition = D_16x16;
    for (int r = 0; r < a->nref; r++) {
        m.ref = r; m.ref_cost = ref_cost(a, r);
        i_halfpel_thresh -= m.ref_cost;
        a->cur_valid = 0;
        predict_mv(a, 0, 0, 2, r, m.mvp);
        if (r == e->blind_dupe) {
            m.mv[0] = a->mvc[0][0][0]; m.mv[1] = a->mvc[0][0][1];
            me_refine_qpel_refdupe(a, &m, p_halfpel_thresh);
        } else {
            const int i_mvc = predict_mv_ref16x16(a, r, mvc);
            me_search_ref(a, &m, mvc, i_mvc, p_halfpel_thresh);
        }
        /* save the vector for predicting neighbours */
        int16_t (*mvr)[2] = r == 0 ? e->mv16[e->cur] : e->mvr[r];
        mvr[a->mi][0] = (int16_t)m.mv[0]; mvr[a->mi][1] = (int16_t)m.mv[1];
        a->mvc[r][0][0] = m.mv[0]; a->mvc[r][0][1] = m.mv[1];
        /* early termination: the skip vector is (almost) the search result and the residual would vanish */
        if (r == 0 && a->b_try_skip && m.cost - m.cost_mv < 300 * a->lambda &&
            abs(m.mv[0] - a->pskip_mv[0]) + abs(m.mv[1] - a->pskip_mv[1]) <= 1 && probe_pskip(a)) return 1;
        m.cost += m.ref_cost;
        i_halfpel_thresh += m.ref_cost;
        if (m.cost < a->me16.cost) a->me16 = m;
    }
    cache_block(a, 0, 0, 2, 2, a->me16.ref, NULL);
    if (a->mbrd && a->me16.ref == 0 && a->me16.mv[0] == a->pskip_mv[0] && a->me16.mv[1] == a->pskip_mv[1]) {
        /* RD: the 16x16 result IS the skip vector — cost it now; nothing coded means P_SKIP */
        x264gpu_mb *mb = &e->mbs[a->mi];
        a->rd16x16 = rd_cost_inter(a, D_16x16, 0, mb, e->levels + (size_t)a->mi * X264GPU_MB_LEVELS);
        if (mb->type == X264GPU_MB_P_SKIP) return 1;
    }
    return 0;
}

static void analyse_inter_p8x8_mixed_ref(actx *a)
{
    int i_maxref = a->nref - 1;
    const int dupe = a->e->blind_dupe;
    a->partition = D_8x8;
    /* early termination: if 16x16 chose reference 0, evaluate no references older than those used by the neighbours
     * (x264 tests the neighbour types with "> 0": unavailable and Intra4x4 neighbours both fail it) */
    if (a->b_early_terminate && i_maxref > 0 && (a->me16.ref == 0 || a->me16.ref == dupe) && a->type_top > 0 && a->type_left > 0) {
        const int gx = 2 * a->mbx, gy = 2 * a->mby;
        const nb_t n[6] = { nb8(a, gx - 1, gy - 1), nb8(a, gx, gy - 1), nb8(a, gx + 1, gy - 1), nb8(a, gx + 2, gy - 1), nb8(a, gx - 1, gy), nb8(a, gx - 1, gy + 1) };
        i_maxref = 0;
        for (int i = 0; i < 6; i++) if (n[i].ref > i_maxref && n[i].ref != dupe) i_maxref = n[i].ref;
    }
    for (int r = 0; r <= i_maxref; r++) {
        const int16_t (*mvr)[2] = r == 0 ? a->e->mv16[a->e->cur] : a->e->mvr[r];
        a->mvc[r][0][0] = mvr[a->mi][0]; a->mvc[r][0][1] = mvr[a->mi][1];
    }
    a->cur_valid = 0;
    for (int i = 0; i < 4; i++) {
        me_t *l0m = &a->me8[i], m;
        const int x8 = i & 1, y8 = i >> 1;
        m.w = m.h = 8; m.ox = 8 * x8; m.oy = 8 * y8; m.list = 0;
        l0m->cost = 0x7fffffff;
        for (int r = 0; r <= i_maxref || r == dupe;) {
            m.ref = r; m.ref_cost = ref_cost(a, r);
            a->cur8[i].ref = r;             /* x264_macroblock_cache_ref before predicting (the block itself is not a neighbour) */
            predict_mv(a, x8, y8, 1, r, m.mvp);
            if (r == dupe) {
                m.mv[0] = a->mvc[0][i + 1][0]; m.mv[1] = a->mvc[0][i + 1][1];
                me_refine_qpel_refdupe(a, &m, NULL);
            } else me_search_ref(a, &m, a->mvc[r], i + 1, NULL);
            m.cost += m.ref_cost;
            a->mvc[r][i + 1][0] = m.mv[0]; a->mvc[r][i + 1][1] = m.mv[1];
            if (m.cost < l0m->cost) *l0m = m;
            /* the duplicate is visited even when the early termination cut the loop short of it */
            if (r == i_maxref && i_maxref < dupe) r = dupe; else r++;
        }
        cache_block(a, x8, y8, 1, 1, l0m->ref, l0m->mv);
        a->satd8x8[i] = l0m->cost - (l0m->cost_mv + l0m->ref_cost);
        /* sub-macroblock type cost: under CABAC without sub-8x8 analysis it is effectively zero */
        if (!a->e->cfg.cabac) l0m->cost += a->lambda * 1;
    }
    a->cost8x8 = a->me8[0].cost + a->me8[1].cost + a->me8[2].cost + a->me8[3].cost;
    /* P_8x8ref0 has no reference cost (CAVLC) */
    if (!a->e->cfg.cabac && !(a->me8[0].ref | a->me8[1].ref | a->me8[2].ref | a->me8[3].ref)) a->cost8x8 -= ref_cost(a, 0) * 4;
}

static void analyse_inter_p8x8(actx *a)
{
    /* duplicates are rarely worth their reference bits in P_8x8: without mixed references they are not analysed (x264) */
    const int r = a->me16.ref == a->e->blind_dupe ? 0 : a->me16.ref;
    const int i_ref_cost = (a->e->cfg.cabac || r) ? ref_cost(a, r) : 0;          /* CAVLC: reference 0 of P_8x8 costs nothing (P_8x8ref0) */
    int i_mvc = 1;
    a->partition = D_8x8;
    a->mvc[r][0][0] = a->me16.mv[0]; a->mvc[r][0][1] = a->me16.mv[1];
    a->cur_valid = 0;
    for (int i = 0; i < 4; i++) {
        me_t *m = &a->me8[i];
        const int x8 = i & 1, y8 = i >> 1;
        m->w = m->h = 8; m->ox = 8 * x8; m->oy = 8 * y8; m->ref = r; m->ref_cost = i_ref_cost;
        predict_mv(a, x8, y8, 1, r, m->mvp);
        me_search_ref(a, m, a->mvc[r], i_mvc, NULL);
        cache_block(a, x8, y8, 1, 1, r, m->mv);
        a->mvc[r][i_mvc][0] = m->mv[0]; a->mvc[r][i_mvc][1] = m->mv[1];
        i_mvc++;
        a->satd8x8[i] = m->cost - m->cost_mv;
        m->cost += i_ref_cost;
        if (!a->e->cfg.cabac) m->cost += a->lambda * 1;
    }
    a->cost8x8 = a->me8[0].cost + a->me8[1].cost + a->me8[2].cost + a->me8[3].cost;
}

static void analyse_inter_p16x8(actx *a, int i_best_satd)
{
    int mvc[3][2];
    a->partition = D_16x8;
    a->cur_valid = 0;
    for (int i = 0; i < 2; i++) {
        me_t *l0m = &a->me16x8[i], m;
        const int r0 = a->me8[2 * i].ref, r1 = a->me8[2 * i + 1].ref;
        const int ref8[2] = { r0 < r1 ? r0 : r1, r0 < r1 ? r1 : r0 };
        const int i_ref8s = ref8[0] == ref8[1] ? 1 : 2;
        m.w = 16; m.h = 8; m.ox = 0; m.oy = 8 * i; m.list = 0;
        l0m->cost = 0x7fffffff;
        for (int j = 0; j < i_ref8s; j++) {
            const int r = ref8[j];
            m.ref = r; m.ref_cost = ref_cost(a, r);
            for (int k = 0; k < 2; k++) { mvc[0][k] = a->mvc[r][0][k]; mvc[1][k] = a->mvc[r][2 * i + 1][k]; mvc[2][k] = a->mvc[r][2 * i + 2][k]; }
            a->cur8[2 * i].ref = a->cur8[2 * i + 1].ref = r;
            predict_mv(a, 0, i, 2, r, m.mvp);
            /* the duplicate right after a search of reference 0: keep that search's vector, refine only */
            if (r == a->e->blind_dupe && !ref8[0]) me_refine_qpel_refdupe(a, &m, NULL);
            else me_search_ref(a, &m, mvc, 3, NULL);
            m.cost += m.ref_cost;
            if (m.cost < l0m->cost) *l0m = m;
        }
        /* early termination on the first half plus the estimate of the second */
        if (a->b_early_terminate && !i && l0m->cost + a->cost_est16x8_1 > i_best_satd * (4 + !!a->mbrd) / 4) { a->cost16x8 = COST_MAX; return; }
        cache_block(a, 0, i, 2, 1, l0m->ref, l0m->mv);
    }
    a->cost16x8 = a->me16x8[0].cost + a->me16x8[1].cost;
}

static void analyse_inter_p8x16(actx *a, int i_best_satd)
{
    int mvc[3][2];
    a->partition = D_8x16;
    a->cur_valid = 0;
    for (int i = 0; i < 2; i++) {
        me_t *l0m = &a->me8x16[i], m;
        const int r0 = a->me8[i].ref, r1 = a->me8[i + 2].ref;
        const int ref8[2] = { r0 < r1 ? r0 : r1, r0 < r1 ? r1 : r0 };
        const int i_ref8s = ref8[0] == ref8[1] ? 1 : 2;
        m.w = 8; m.h = 16; m.ox = 8 * i; m.oy = 0; m.list = 0;
        l0m->cost = 0x7fffffff;
        for (int j = 0; j < i_ref8s; j++) {
            const int r = ref8[j];
            m.ref = r; m.ref_cost = ref_cost(a, r);
            for (int k = 0; k < 2; k++) { mvc[0][k] = a->mvc[r][0][k]; mvc[1][k] = a->mvc[r][i + 1][k]; mvc[2][k] = a->mvc[r][i + 3][k]; }
            a->cur8[i].ref = a->cur8[i + 2].ref = r;
            predict_mv(a, i, 0, 1, r, m.mvp);
            if (r == a->e->blind_dupe && !ref8[0]) me_refine_qpel_refdupe(a, &m, NULL);
            else me_search_ref(a, &m, mvc, 3, NULL);
            m.cost += m.ref_cost;
            if (m.cost < l0m->cost) *l0m = m;
        }
        if (a->b_early_terminate && !i && l0m->cost + a->cost_est8x16_1 > i_best_satd * (4 + !!a->mbrd) / 4) { a->cost8x16 = COST_MAX; return; }
        cache_block(a, i, 0, 1, 2, l0m->ref, l0m->mv);
    }
    a->cost8x16 = a->me8x16[0].cost + a->me8x16[1].cost;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * intra analysis (x264_mb_analyse_intra, x264_mb_analyse_intra_chroma) on the reconstruction of the coded neighbours.
 * The block loops code each block as they go (the next block predicts from it); the luma reconstruction is restored afterwards:
 * x264_macroblock_encode codes the chosen type again from scratch. */
static int i4_pred_mode(const actx *a, int b, const uint8_t *cur_modes)
{
    /* 8.3.1.1 / x264_mb_predict_intra4x4_mode: min of the left / top block modes; DC when a neighbour is absent; neighbour macroblocks that are
     * not I_NxN count as DC.  I8x8 macroblocks store each 8x8 mode replicated over their 4x4 entries. */
    const x264o_encoder *e = a->e;
    const int bx = blk_x[b], by = blk_y[b];
    int ma, mb_;
    if (bx > 0) ma = cur_modes[idx_of[by][bx - 1]];
    else if (a->mbx > 0) { const x264gpu_mb *n = &e->mbs[a->mi - 1]; ma = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[idx_of[by][3]] : 2; }
    else return 2;
    if (by > 0) mb_ = cur_modes[idx_of[by - 1][bx]];
    else if (a->mby > e->row0) { const x264gpu_mb *n = &e->mbs[a->mi - e->mbw]; mb_ = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[idx_of[3][bx]] : 2; }
    else return 2;
    return ma < mb_ ? ma : mb_;
}
static int i4_avail(const actx *a, int b)
{
    const int bx = blk_x[b], by = blk_y[b], mbx = a->mbx, mby = a->mby - a->e->row0;      /* row within the slice */
    int av = 0;
    if (bx > 0 || mbx > 0) av |= X264O_AVAIL_LEFT;
    if (by > 0 || mby > 0) av |= X264O_AVAIL_TOP;
    if ((bx > 0 || mbx > 0) && (by > 0 || mby > 0)) av |= X264O_AVAIL_TOPLEFT;
    if (by == 0) { if (mby > 0 && (bx < 3 || mbx + 1 < a->e->mbw)) av |= X264O_AVAIL_TOPRIGHT; }
    else if (bx < 3 && idx_of[by - 1][bx + 1] < b) av |= X264O_AVAIL_TOPRIGHT;
    return av;
}
static int i8_avail(const actx *a, int i8)
{
    const int x8 = i8 & 1, y8 = i8 >> 1, left = a->mbx > 0, top = a->mby > a->e->row0;
    int av = 0;
    if (x8 || left) av |= X264O_AVAIL_LEFT;
    if (y8 || top) av |= X264O_AVAIL_TOP;
    if ((x8 || left) && (y8 || top)) av |= X264O_AVAIL_TOPLEFT;
    if (i8 == 0 ? top : i8 == 1 ? (top && a->mbx + 1 < a->e->mbw) : i8 == 2) av |= X264O_AVAIL_TOPRIGHT;
    return av;
}
static int real_mode4(int m, int avail)       /* DC variant by availability */
{
    if (m != I_PRED_4x4_DC) return m;
    const int l = avail & X264O_AVAIL_LEFT, t = avail & X264O_AVAIL_TOP;
    return l && t ? I_PRED_4x4_DC : l ? I_PRED_4x4_DC_LEFT : t ? I_PRED_4x4_DC_TOP : I_PRED_4x4_DC_128;
}
/* predict_4x4_mode_available / i4x4_mode_available: candidate list by neighbour availability */
static const int8_t *mode4_available(int avail)
{
    static const int8_t tab[5][10] = {
        { I_PRED_4x4_DC, -1, -1, -1, -1, -1, -1, -1, -1, -1 },
        { I_PRED_4x4_DC, I_PRED_4x4_H, I_PRED_4x4_HU, -1, -1, -1, -1, -1, -1, -1 },
        { I_PRED_4x4_DC, I_PRED_4x4_V, I_PRED_4x4_DDL, I_PRED_4x4_VL, -1, -1, -1, -1, -1, -1 },
        { I_PRED_4x4_DC, I_PRED_4x4_H, I_PRED_4x4_V, I_PRED_4x4_DDL, I_PRED_4x4_VL, I_PRED_4x4_HU, -1, -1, -1, -1 },
        { I_PRED_4x4_V, I_PRED_4x4_H, I_PRED_4x4_DC, I_PRED_4x4_DDL, I_PRED_4x4_DDR, I_PRED_4x4_VR, I_PRED_4x4_HD, I_PRED_4x4_VL, I_PRED_4x4_HU, -1 } };
    const int all = X264O_AVAIL_LEFT | X264O_AVAIL_TOP | X264O_AVAIL_TOPLEFT;
    return tab[(avail & all) == all ? 4 : avail & (X264O_AVAIL_LEFT | X264O_AVAIL_TOP)];
}
/* intra_analysis_shortcut[all nine available][favor_vertical]: directional modes still worth trying after V / H / DC */
static const int8_t intra_analysis_shortcut[2][2][5] = {
    { { I_PRED_4x4_HU, -1, -1, -1, -1 }, { I_PRED_4x4_DDL, I_PRED_4x4_VL, -1, -1, -1 } },
    { { I_PRED_4x4_DDR, I_PRED_4x4_HD, I_PRED_4x4_HU, -1, -1 }, { I_PRED_4x4_DDL, I_PRED_4x4_DDR, I_PRED_4x4_VR, I_PRED_4x4_VL, -1 } } };

static void analyse_intra_chroma(actx *a)
{
    if (a->satd_chroma < COST_MAX) return;
    x264o_encoder *e = a->e;
    const int left = a->mbx > 0, top = a->mby > a->e->row0;
    const pixel *fuv = e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16;
    const pixel *ruv = chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    pixel fu[64], fv[64], nu[9 * 9], nvv[9 * 9], pu[64], pv[64];
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    memset(nu, 128, sizeof(nu)); memset(nvv, 128, sizeof(nvv));
    for (int y = -1; y < 8; y++)
        for (int x = -1; x < 8; x++) {
            if (y >= 0 && x >= 0) continue;
            if ((y < 0 && !top) || (x < 0 && !left)) continue;
            nu[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x]; nvv[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x + 1];
        }
    int cm[4], cn = 0;
    if (left && top) { cm[cn++] = I_PRED_CHROMA_DC; cm[cn++] = I_PRED_CHROMA_H; cm[cn++] = I_PRED_CHROMA_V; cm[cn++] = I_PRED_CHROMA_P; }
    else if (left) { cm[cn++] = I_PRED_CHROMA_DC_LEFT; cm[cn++] = I_PRED_CHROMA_H; }
    else if (top) { cm[cn++] = I_PRED_CHROMA_DC_TOP; cm[cn++] = I_PRED_CHROMA_V; }
    else cm[cn++] = I_PRED_CHROMA_DC_128;
    for (int i = 0; i < cn; i++) {
        const int m = cm[i], sig = m > I_PRED_CHROMA_P ? I_PRED_CHROMA_DC : m;
        x264o_predict_8x8c(pu, 8, nu + 10, 9, m);
        x264o_predict_8x8c(pv, 8, nvv + 10, 9, m);
        const int c = mbcmp(a, pu, 8, fu, 8, 8, 8) + mbcmp(a, pv, 8, fv, 8, 8, 8) + a->lambda * bs_size_ue(sig);
        a->satd_chroma_dir[m] = c;
        if (c < a->satd_chroma) { a->satd_chroma = c; a->predc = m; }
    }
}

static void analyse_intra(actx *a, int i_satd_inter)
{
    x264o_encoder *e = a->e;
    const int lambda = a->lambda, qp = a->qp, left = a->mbx > 0, top = a->mby > a->e->row0;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    const int parts = (e->slice_type == X264GPU_SLICE_I && (e->cfg.partitions & 0x100)) ? (e->cfg.partitions >> 8) & 6 : e->cfg.partitions & 7;
    /* B slices: the macroblock type prefix of an intra type enters its SATD cost (x264 i_mb_b_cost_table[I_4x4 / I_8x8 / I_16x16] = 9) */
    const int b_type_cost = e->slice_type == X264GPU_SLICE_B ? 9 * lambda : 0;
    pixel save[256], pred[256];
    for (int y = 0; y < 16; y++) memcpy(save + y * 16, rec + y * e->rs, 16);
    /* ---- 16x16: V, H, DC first; plane only if one of them was useful ---- */
    {
        static const uint8_t i16x16_thresh_lut[11] = { 2, 2, 2, 3, 3, 4, 4, 4, 4, 4, 4 };
        const int i16x16_thresh = a->b_fast_intra ? (i16x16_thresh_lut[a->subme > 10 ? 10 : a->subme] * i_satd_inter) >> 1 : COST_MAX;
        if (left && top) {
            for (int m = 0; m < 3; m++) {       /* V, H, DC */
                x264o_predict_16x16(pred, 16, rec, e->rs, m);
                const int c = mbcmp(a, pred, 16, fenc, e->fs, 16, 16) + lambda * bs_size_ue(m);
                a->satd_i16_dir[m] = c;
                if (c < a->satd_i16) { a->satd_i16 = c; a->pred16 = m; }
            }
            if (a->satd_i16 <= i16x16_thresh) {
                x264o_predict_16x16(pred, 16, rec, e->rs, I_PRED_16x16_P);
                const int c = mbcmp(a, pred, 16, fenc, e->fs, 16, 16) + lambda * bs_size_ue(3);
                a->satd_i16_dir[I_PRED_16x16_P] = c;
                if (c < a->satd_i16) { a->satd_i16 = c; a->pred16 = I_PRED_16x16_P; }
            }
        } else {
            int modes[2], n = 0;
            if (left) { modes[n++] = I_PRED_16x16_DC_LEFT; modes[n++] = I_PRED_16x16_H; }
            else if (top) { modes[n++] = I_PRED_16x16_DC_TOP; modes[n++] = I_PRED_16x16_V; }
            else modes[n++] = I_PRED_16x16_DC_128;
            for (int i = 0; i < n; i++) {
                const int m = modes[i], sig = m > I_PRED_16x16_P ? I_PRED_16x16_DC : m;
                x264o_predict_16x16(pred, 16, rec, e->rs, m);
                const int c = mbcmp(a, pred, 16, fenc, e->fs, 16, 16) + lambda * bs_size_ue(sig);
                a->satd_i16_dir[m] = c;
                if (c < a->satd_i16) { a->satd_i16 = c; a->pred16 = m; }
            }
        }
        a->satd_i16 += b_type_cost;
        if (a->satd_i16 > i16x16_thresh) return;
    }
    /* ---- 8x8 ---- */
    if ((parts & 4) && e->cfg.dct8x8) {
        /* under RD every 8x8 block is analysed: the RD cost decides, not the running SATD sum */
        const int i_satd_thresh = a->mbrd ? COST_MAX : i_satd_inter < a->satd_i16 ? i_satd_inter : a->satd_i16;
        int i_cost = lambda * 4 + b_type_cost, idx;
        uint8_t m8[16];
        int16_t lvtmp[256]; uint32_t nnztmp = 0;
        memset(m8, 2, sizeof(m8));
        for (idx = 0;; idx++) {
            const int x8 = idx & 1, y8 = idx >> 1, avail = i8_avail(a, idx);
            const pixel *f = fenc + y8 * 8 * e->fs + x8 * 8;
            pixel *r = rec + y8 * 8 * e->rs + x8 * 8;
            const int i_pred_mode = i4_pred_mode(a, idx * 4, m8);
            const int8_t *predict_mode = mode4_available(avail);
            pixel edge[33], p8[64];
            int i_best = COST_MAX, bestm = 2;
            x264o_predict_8x8_filter(r, e->rs, edge, avail);
            if (predict_mode[5] >= 0) {
                int satd[9];
                for (int m = 0; m < 3; m++) { x264o_predict_8x8(p8, 8, edge, m); satd[m] = a->satd ? x264o_sa8d(p8, 8, f, e->fs, 8, 8) : x264o_sad(p8, 8, f, e->fs, 8, 8); }
                const int favor_vertical = satd[I_PRED_4x4_H] > satd[I_PRED_4x4_V];
                if (i_pred_mode < 3) satd[i_pred_mode] -= 3 * lambda;
                for (int i = 2; i >= 0; i--) { a->satd_i8_dir[idx][i] = satd[i] + 4 * lambda; if (satd[i] < i_best) { i_best = satd[i]; bestm = i; } }
                /* analysis shortcut: skip the modes far from the favoured direction — unless RD decides and fast-intra is off (i_mbrd < 1 + b_fast_intra) */
                if (a->mbrd < 1 + a->b_fast_intra) predict_mode = intra_analysis_shortcut[predict_mode[8] >= 0][favor_vertical];
                else predict_mode += 3;
            }
            for (; *predict_mode >= 0 && (i_best >= 0 || a->mbrd >= 2); predict_mode++) {       /* (RD refinement wants every mode's cost) */
                const int m = *predict_mode;
                x264o_predict_8x8(p8, 8, edge, real_mode4(m, avail));
                int c = a->satd ? x264o_sa8d(p8, 8, f, e->fs, 8, 8) : x264o_sad(p8, 8, f, e->fs, 8, 8);
                if (i_pred_mode == m) c -= 3 * lambda;
                if (c < i_best) { i_best = c; bestm = m; }
                a->satd_i8_dir[idx][m] = c + 4 * lambda;
            }
            i_cost += i_best + 3 * lambda;
            a->pred8[idx] = bestm;
            memset(m8 + idx * 4, bestm, 4);
            if (idx == 3 || i_cost > i_satd_thresh) break;
            /* code the block: the next ones predict from it */
            x264o_predict_8x8(p8, 8, edge, real_mode4(bestm, avail));
            for (int y = 0; y < 8; y++) memcpy(r + y * e->rs, p8 + y * 8, 8);
            encode_i8x8(e, f, r, qp, idx, lvtmp, &nnztmp);
        }
        if (idx == 3) a->satd_i8 = i_cost;
        else {
            static const uint16_t cost_div_fix8[3] = { 1024, 512, 341 };
            a->satd_i8 = COST_MAX;
            i_cost = (i_cost * cost_div_fix8[idx]) >> 8;
        }
        for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, save + y * 16, 16);
        static const uint8_t i8x8_thresh[11] = { 4, 4, 4, 5, 5, 5, 6, 6, 6, 6, 6 };
        const int mn = i_cost < a->satd_i16 ? i_cost : a->satd_i16;
        if (a->b_early_terminate && mn > (int)(((int64_t)i_satd_inter * i8x8_thresh[a->subme > 10 ? 10 : a->subme]) >> 2)) return;
    }
    /* ---- 4x4 ---- */
    if (parts & 2) {
        int i_cost = lambda * (24 + 16) + b_type_cost, idx;
        int i_satd_thresh = COST_MAX;
        if (a->b_early_terminate) { i_satd_thresh = i_satd_inter < a->satd_i16 ? i_satd_inter : a->satd_i16; if (a->satd_i8 < i_satd_thresh) i_satd_thresh = a->satd_i8; }
        if (a->b_early_terminate && a->mbrd) i_satd_thresh = (int)((int64_t)i_satd_thresh * (10 - a->b_fast_intra) / 8);      /* RD: a little slack, the SATD order is not final */
        uint8_t m4[16];
        int16_t l16[16];
        memset(m4, 2, sizeof(m4));
        for (idx = 0;; idx++) {
            const pixel *f = fenc + blk_y[idx] * 4 * e->fs + blk_x[idx] * 4;
            pixel *r = rec + blk_y[idx] * 4 * e->rs + blk_x[idx] * 4;
            const int avail = i4_avail(a, idx), i_pred_mode = i4_pred_mode(a, idx, m4);
            const int8_t *predict_mode = mode4_available(avail);
            pixel p4[16];
            int i_best = COST_MAX, bestm = 2;
            if (predict_mode[5] >= 0) {
                int satd[3];
                for (int m = 0; m < 3; m++) { x264o_predict_4x4(p4, 4, r, e->rs, m, avail); satd[m] = mbcmp(a, p4, 4, f, e->fs, 4, 4); }
                const int favor_vertical = satd[I_PRED_4x4_H] > satd[I_PRED_4x4_V];
                if (i_pred_mode < 3) satd[i_pred_mode] -= 3 * lambda;
                i_best = satd[I_PRED_4x4_DC]; bestm = I_PRED_4x4_DC;
                if (satd[I_PRED_4x4_H] < i_best) { i_best = satd[I_PRED_4x4_H]; bestm = I_PRED_4x4_H; }
                if (satd[I_PRED_4x4_V] < i_best) { i_best = satd[I_PRED_4x4_V]; bestm = I_PRED_4x4_V; }
                if (a->mbrd < 1 + a->b_fast_intra) predict_mode = intra_analysis_shortcut[predict_mode[8] >= 0][favor_vertical];
                else predict_mode += 3;
            }
            if (i_best > 0)
                for (; *predict_mode >= 0; predict_mode++) {
                    const int m = *predict_mode;
                    x264o_predict_4x4(p4, 4, r, e->rs, real_mode4(m, avail), avail);
                    int c = mbcmp(a, p4, 4, f, e->fs, 4, 4);
                    if (i_pred_mode == m) {
                        c -= lambda * 3;
                        if (c <= 0) { i_best = c; bestm = m; break; }
                    }
                    if (c < i_best) { i_best = c; bestm = m; }
                }
            i_cost += i_best + 3 * lambda;
            a->pred4[idx] = bestm;
            m4[idx] = (uint8_t)bestm;
            if (i_cost > i_satd_thresh || idx == 15) break;
            x264o_predict_4x4(p4, 4, r, e->rs, real_mode4(bestm, avail), avail);
            for (int y = 0; y < 4; y++) memcpy(r + y * e->rs, p4 + y * 4, 4);
            encode_i4x4(e, f, r, qp, l16);
        }
        a->satd_i4 = idx == 15 ? i_cost : COST_MAX;
        for (int y = 0; y < 16; y++) memcpy(rec + y * e->rs, save + y * 16, 16);
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * x264_macroblock_encode */
static void encode_intra_chroma(actx *a, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int left = a->mbx > 0, top = a->mby > a->e->row0;
    const pixel *fuv = e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16;
    pixel *ruv = chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    pixel nu[9 * 9], nvv[9 * 9], pu[64], pv[64];
    memset(nu, 128, sizeof(nu)); memset(nvv, 128, sizeof(nvv));
    for (int y = -1; y < 8; y++)
        for (int x = -1; x < 8; x++) {
            if (y >= 0 && x >= 0) continue;
            if ((y < 0 && !top) || (x < 0 && !left)) continue;
            nu[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x]; nvv[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x + 1];
        }
    x264o_predict_8x8c(pu, 8, nu + 10, 9, a->predc);
    x264o_predict_8x8c(pv, 8, nvv + 10, 9, a->predc);
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { ruv[y * e->rs + 2 * x] = pu[y * 8 + x]; ruv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
    mb->chroma_mode = (uint8_t)(a->predc > I_PRED_CHROMA_P ? I_PRED_CHROMA_DC : a->predc);
    encode_chroma(e, fuv, ruv, a->qpc, 0, mb, lv);
}

static void encode_intra_mb(actx *a, int type, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    mb->type = (uint8_t)type;
    for (int k = 0; k < 4; k++) mb->ref[k] = -1;
    if (type == X264GPU_MB_I16x16) {
        mb->i16_mode = (uint8_t)(a->pred16 > I_PRED_16x16_P ? I_PRED_16x16_DC : a->pred16);
        encode_i16x16(e, fenc, rec, a->qp, a->pred16, mb, lv);
    } else if (type == X264GPU_MB_I8x8) {
        mb->transform8x8 = 1;
        for (int i8 = 0; i8 < 4; i8++) {
            const pixel *f = fenc + (i8 >> 1) * 8 * e->fs + (i8 & 1) * 8;
            pixel *r = rec + (i8 >> 1) * 8 * e->rs + (i8 & 1) * 8;
            const int avail = i8_avail(a, i8);
            pixel edge[33], p8[64];
            memset(mb->i4_mode + i8 * 4, a->pred8[i8], 4);
            x264o_predict_8x8_filter(r, e->rs, edge, avail);
            x264o_predict_8x8(p8, 8, edge, real_mode4(a->pred8[i8], avail));
            for (int y = 0; y < 8; y++) memcpy(r + y * e->rs, p8 + y * 8, 8);
            if (encode_i8x8(e, f, r, a->qp, i8, lv, &mb->nnz)) mb->cbp_luma |= 1 << i8;
        }
    } else {
        for (int b = 0; b < 16; b++) {
            const pixel *f = fenc + blk_y[b] * 4 * e->fs + blk_x[b] * 4;
            pixel *r = rec + blk_y[b] * 4 * e->rs + blk_x[b] * 4;
            const int avail = i4_avail(a, b);
            pixel p4[16];
            mb->i4_mode[b] = (uint8_t)a->pred4[b];
            x264o_predict_4x4(p4, 4, r, e->rs, real_mode4(a->pred4[b], avail), avail);
            for (int y = 0; y < 4; y++) memcpy(r + y * e->rs, p4 + y * 4, 4);
            if (encode_i4x4(e, f, r, a->qp, lv + b * 16)) { mb->nnz |= 1u << b; mb->cbp_luma |= 1 << (b >> 2); }
        }
    }
    analyse_intra_chroma(a);          /* x264_analyse_update_cache: chroma mode of intra macroblocks, unless chroma-ME already did it */
    encode_intra_chroma(a, mb, lv);
}

static void encode_inter_mb(actx *a, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int mbx = a->mbx, mby = a->mby;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)mby * 16 * e->rs + mbx * 16;
    pixel *rec_uv = chroma_plane(e, e->cur) + (size_t)mby * 8 * e->rs + mbx * 16;
    const pixel *fenc = e->fenc_y + (size_t)mby * 16 * e->fs + mbx * 16;
    pixel pu[64], pv[64];
    if (mb->type >= X264GPU_MB_B_DIRECT) mc_mb_b(e, mbx, mby, mb, rec, e->rs, pu, pv);
    else
    for (int k = 0; k < 4; k++)       /* motion compensation per 8x8 quadrant (covers 16x16 / 16x8 / 8x16 / 8x8) */
        mc_mb_p(e, mbx, mby, (k & 1) * 8, (k >> 1) * 8, 8, 8, mb->ref[k], mb->mv[k][0], mb->mv[k][1], rec, e->rs, pu, pv, 8);
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) { rec_uv[y * e->rs + 2 * x] = pu[y * 8 + x]; rec_uv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
    if (mb->type == X264GPU_MB_P_SKIP || mb->type == X264GPU_MB_B_SKIP) return;                /* x264_macroblock_encode_skip: the prediction is the reconstruction */
    /* x264_mb_analyse_transform: SA8D vs SATD of the prediction error */
    mb->transform8x8 = 0;
    if (a->force_t8 >= 0) mb->transform8x8 = (uint8_t)a->force_t8;          /* RD: the transform size is a candidate property (x264_mb_analyse_transform_rd) */
    else if (e->cfg.dct8x8) mb->transform8x8 = x264o_sa8d(fenc, e->fs, rec, e->rs, 16, 16) < x264o_satd(fenc, e->fs, rec, e->rs, 16, 16);
    if (mb->transform8x8) encode_luma_inter8(e, fenc, rec, a->qp, mb, lv);
    else encode_luma_inter(e, fenc, rec, a->qp, mb, lv);
    if (!mb->cbp_luma) mb->transform8x8 = 0;      /* the flag is not transmitted without luma coefficients (macroblock_cache_save) */
    encode_chroma(e, e->fenc_uv + (size_t)mby * 8 * e->fs + mbx * 16, rec_uv, a->qpc, 1, mb, lv);
    /* P_L0 16x16, reference 0, skip vector, nothing coded: P_SKIP */
    if (mb->type == X264GPU_MB_P_L0 && mb->partition == D_16x16 && !(mb->cbp_luma | mb->cbp_chroma) && mb->ref[0] == 0 &&
        mb->mv[0][0] == a->pskip_mv[0] && mb->mv[0][1] == a->pskip_mv[1]) mb->type = X264GPU_MB_P_SKIP;
    /* B_DIRECT with nothing coded: B_SKIP */
    if (mb->type == X264GPU_MB_B_DIRECT && !(mb->cbp_luma | mb->cbp_chroma)) mb->type = X264GPU_MB_B_SKIP;
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * x264_macroblock_analyse + x264_macroblock_encode for macroblock (mbx,mby) */

/* ===========================================================================================================================
 * Rate-distortion mode decision (cfg.rd: x264 subme 6 / 7, i_mbrd 1; [x264-upstream] encoder/rdo.c x264_rd_cost_mb, encoder/analyse.c
 * x264_mb_analyse_p_rd / _transform_rd / x264_intra_rd), with the bit counts of a CAVLC session (x264_macroblock_size_cavlc: exactly the
 * bits the macroblock layer would take, without the skip run).  Restated from memory like the rest of this file. */
int x264o_cavlc_block_bits(const int16_t *l, int n, int nC, int *total_out);       /* rd.cpp */
int x264o_cavlc_cbp_bits(int cbp, int inter);
static int bs_size_se(int v) { return bs_size_ue(v <= 0 ? -2 * v : 2 * v - 1); }

/* total_coeff of block b (0..15 luma, 16..23 chroma AC: plane * 4 + block) of a finished macroblock, as the nC derivation sees it */
static int nb_total_coeff(const x264o_encoder *e, int mi, int b)
{
    const x264gpu_mb *m = &e->mbs[mi];
    const int16_t *lv = e->levels + (size_t)mi * X264GPU_MB_LEVELS;
    int c = 0;
    if (m->type == X264GPU_MB_P_SKIP) return 0;
    if (b < 16) {
        if (!((m->cbp_luma >> (b >> 2)) & 1)) return 0;
        for (int i = m->type == X264GPU_MB_I16x16; i < 16; i++) c += lv[b * 16 + i] != 0;
    } else {
        if (m->cbp_chroma != 2) return 0;
        for (int i = 1; i < 16; i++) c += lv[X264GPU_LV_CHROMA_AC + (b - 16) * 16 + i] != 0;
    }
    return c;
}

/* bits of the macroblock layer of *mb (already coded into lv; neighbours from e->mbs) in a CAVLC slice */
static void cache_block_l(actx *a, int l, int bx8, int by8, int w8, int h8, int ref, const int mv[2]);
static int mb_bits_cavlc(actx *a, const x264gpu_mb *mb, const int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int pslice = e->slice_type == X264GPU_SLICE_P, left = a->mbx > 0, top = a->mby > e->row0, mi = a->mi;
    const int intra = mb->type <= X264GPU_MB_I16x16;
    int bits = 0, tc[24];
    memset(tc, 0, sizeof(tc));
    const int bsl = e->slice_type == X264GPU_SLICE_B;
    if (!intra && bsl) {
        /* cavlc_mb_header_b ([x264-upstream] encoder/cavlc.c) as a count: mb_type ue(v) of Table 7-14, the four sub_mb_type of B_8x8 (direct 0, L0 1, L1 2,
         * Bi 3), te(v) reference indices of list 0 then list 1, vector differences of list 0 then list 1, each over the partitions that use the list */
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        static const int8_t pair_of[3][3] = { { 0, 2, 4 }, { 3, 1, 5 }, { 6, 7, 8 } };
        if (mb->type == X264GPU_MB_B_DIRECT) bits += bs_size_ue(0);
        else {
            const int part = mb->partition & 3, nparts = part == D_16x16 ? 1 : part == D_8x8 ? 4 : 2;
            int use[4];
            for (int k = 0; k < nparts; k++) {
                const int b8 = geom[part][k][1] * 2 + geom[part][k][0];
                use[k] = (mb->type == X264GPU_MB_B_8x8 && ((mb->direct8 >> b8) & 1)) ? 3 : mb->ref[b8] >= 0 ? (mb->ref1[b8] >= 0 ? 2 : 0) : 1;
            }
            if (part == D_8x8) { bits += bs_size_ue(22); for (int k = 0; k < 4; k++) bits += bs_size_ue(use[k] == 3 ? 0 : 1 + use[k]); }
            else if (part == D_16x16) bits += bs_size_ue(1 + use[0]);
            else bits += bs_size_ue(4 + 2 * pair_of[use[0]][use[1]] + (part == D_8x16));
            for (int l = 0; l < 2; l++) {
                if (a->nref_l[l] <= 1) continue;
                for (int k = 0; k < nparts; k++) {
                    if (use[k] == 3 || use[k] == 1 - l) continue;
                    const int b8 = geom[part][k][1] * 2 + geom[part][k][0], r = l ? mb->ref1[b8] : mb->ref[b8];
                    bits += a->nref_l[l] == 2 ? 1 : bs_size_ue(r);
                }
            }
            const int part_bak = a->partition, valid_bak = a->cur_valid, valid1_bak = a->cur_valid1;
            nb_t cur_bak[4], cur1_bak[4];
            memcpy(cur_bak, a->cur8, sizeof(cur_bak)); memcpy(cur1_bak, a->cur8b, sizeof(cur1_bak));
            a->partition = part;
            for (int l = 0; l < 2; l++) {
                if (l) a->cur_valid1 = 0; else a->cur_valid = 0;
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[part][k];
                    const int b8 = g[1] * 2 + g[0];
                    const int r = l ? mb->ref1[b8] : mb->ref[b8];
                    const int mv[2] = { l ? mb->mv1[b8][0] : mb->mv[b8][0], l ? mb->mv1[b8][1] : mb->mv[b8][1] };
                    if (!(use[k] == 3 || use[k] == 1 - l)) {
                        int mvp[2];
                        nb_t *cur = l ? a->cur8b : a->cur8;
                        for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) cur[y * 2 + x].ref = r;          /* cache_ref before predicting */
                        predict_mv_l(a, l, g[0], g[1], g[2], r, mvp);
                        bits += bs_size_se(mv[0] - mvp[0]) + bs_size_se(mv[1] - mvp[1]);
                    }
                    cache_block_l(a, l, g[0], g[1], g[2], g[3], r >= 0 ? r : -1, r >= 0 ? mv : NULL);          /* (a direct block's motion is in the record) */
                }
            }
            a->partition = part_bak; a->cur_valid = valid_bak; a->cur_valid1 = valid1_bak; memcpy(a->cur8, cur_bak, sizeof(cur_bak)); memcpy(a->cur8b, cur1_bak, sizeof(cur1_bak));
        }
        bits += x264o_cavlc_cbp_bits(mb->cbp_luma | (mb->cbp_chroma << 4), 1);
        if (e->cfg.dct8x8 && mb->cbp_luma) bits += 1;                               /* transform_size_8x8_flag (direct_8x8_inference is on) */
    } else
    if (!intra) {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        const int nparts = mb->partition == D_16x16 ? 1 : mb->partition == D_8x8 ? 4 : 2;
        const int part_bak = a->partition, valid_bak = a->cur_valid;
        nb_t cur_bak[4];
        memcpy(cur_bak, a->cur8, sizeof(cur_bak));
        bits += bs_size_ue(mb->partition);
        if (mb->partition == D_8x8) bits += 4;                                      /* four sub_mb_type ue(0) */
        if (a->nref > 1) for (int k = 0; k < nparts; k++) { const int8_t *g = geom[mb->partition][k]; bits += a->nref == 2 ? 1 : bs_size_ue(mb->ref[g[1] * 2 + g[0]]); }
        a->partition = mb->partition; a->cur_valid = 0;
        for (int k = 0; k < nparts; k++) {
            const int8_t *g = geom[mb->partition][k];
            const int b8 = g[1] * 2 + g[0];
            int mvp[2];
            const int mv[2] = { mb->mv[b8][0], mb->mv[b8][1] };
            for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) a->cur8[y * 2 + x].ref = mb->ref[b8];   /* cache_ref before predicting */
            predict_mv(a, g[0], g[1], g[2], mb->ref[b8], mvp);
            bits += bs_size_se(mv[0] - mvp[0]) + bs_size_se(mv[1] - mvp[1]);
            cache_block(a, g[0], g[1], g[2], g[3], mb->ref[b8], mv);
        }
        a->partition = part_bak; a->cur_valid = valid_bak; memcpy(a->cur8, cur_bak, sizeof(cur_bak));
        bits += x264o_cavlc_cbp_bits(mb->cbp_luma | (mb->cbp_chroma << 4), 1);
        if (e->cfg.dct8x8 && mb->cbp_luma) bits += 1;                               /* transform_size_8x8_flag */
    } else {
        const int off = bsl ? 23 : pslice ? 5 : 0;
        if (mb->type == X264GPU_MB_I16x16) bits += bs_size_ue(off + 1 + mb->i16_mode + 4 * mb->cbp_chroma + (mb->cbp_luma ? 12 : 0));
        else {
            bits += bs_size_ue(off);
            if (e->cfg.dct8x8) bits += 1;
            if (mb->type == X264GPU_MB_I8x8) { for (int i8 = 0; i8 < 4; i8++) bits += i4_pred_mode(a, i8 * 4, mb->i4_mode) == mb->i4_mode[i8 * 4] ? 1 : 4; }
            else for (int b = 0; b < 16; b++) bits += i4_pred_mode(a, b, mb->i4_mode) == mb->i4_mode[b] ? 1 : 4;
        }
        bits += bs_size_ue(mb->chroma_mode);
        if (mb->type != X264GPU_MB_I16x16) bits += x264o_cavlc_cbp_bits(mb->cbp_luma | (mb->cbp_chroma << 4), 0);
    }
    if (mb->cbp_luma || mb->cbp_chroma || mb->type == X264GPU_MB_I16x16) bits += bs_size_se((int)mb->qp - e->last_qp);
#define NC_OF(na, nb) ((na) >= 0 && (nb) >= 0 ? ((na) + (nb) + 1) >> 1 : (na) >= 0 ? (na) : (nb) >= 0 ? (nb) : 0)
    if (mb->type == X264GPU_MB_I16x16) {
        const int na = left ? nb_total_coeff(e, mi - 1, idx_of[0][3]) : -1, nb = top ? nb_total_coeff(e, mi - e->mbw, idx_of[3][0]) : -1;
        bits += x264o_cavlc_block_bits(lv + X264GPU_LV_LUMA_DC, 16, NC_OF(na, nb), NULL);
    }
    for (int b = 0; b < 16; b++) {
        if (!((mb->cbp_luma >> (b >> 2)) & 1)) continue;
        const int bx = blk_x[b], by = blk_y[b];
        const int na = bx > 0 ? tc[idx_of[by][bx - 1]] : left ? nb_total_coeff(e, mi - 1, idx_of[by][3]) : -1;
        const int nb = by > 0 ? tc[idx_of[by - 1][bx]] : top ? nb_total_coeff(e, mi - e->mbw, idx_of[3][bx]) : -1;
        if (mb->type == X264GPU_MB_I16x16) bits += x264o_cavlc_block_bits(lv + b * 16 + 1, 15, NC_OF(na, nb), &tc[b]);
        else bits += x264o_cavlc_block_bits(lv + b * 16, 16, NC_OF(na, nb), &tc[b]);
    }
    if (mb->cbp_chroma) {
        for (int c = 0; c < 2; c++) bits += x264o_cavlc_block_bits(lv + X264GPU_LV_CHROMA_DC + c * 4, 4, -1, NULL);
        if (mb->cbp_chroma == 2)
            for (int c = 0; c < 2; c++)
                for (int i = 0; i < 4; i++) {
                    const int bx = i & 1, by = i >> 1, base = 16 + c * 4;
                    const int na = bx > 0 ? tc[base + by * 2] : left ? nb_total_coeff(e, mi - 1, base + by * 2 + 1) : -1;
                    const int nb = by > 0 ? tc[base + bx] : top ? nb_total_coeff(e, mi - e->mbw, base + 2 + bx) : -1;
                    bits += x264o_cavlc_block_bits(lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16 + 1, 15, NC_OF(na, nb), &tc[base + i]);
                }
    }
#undef NC_OF
    return bits;
}

/* ssd_mb: luma SSD + the psy-rd energy term, chroma SSD scaled by the chroma lambda offset */
static int rd_ssd_mb(const actx *a)
{
    const x264o_encoder *e = a->e;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16, *fuv = e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16;
    const pixel *rec = luma_plane((x264o_encoder *)e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    const pixel *ruv = chroma_plane((x264o_encoder *)e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    int ssd = x264o_ssd(fenc, e->fs, rec, e->rs, 16, 16);
    if (e->cfg.psy_rd_q8) {
        /* ssd_plane: every size from 8x8 up (here the 16x16 luma) compares pixel_hadamard_ac of reconstruction and source — the AC sums of
         * the 4x4 and of the 8x8 Hadamard transforms, (|d sum4| + |d sum8|) >> 1; only blocks smaller than 8x8 use SATD - SAD / 2
         * (oracle/PSY_NOTES.md) */
        const uint64_t fdec_acs = x264o_hadamard_ac(rec, e->rs, 16, 16), fenc_acs = x264o_hadamard_ac(fenc, e->fs, 16, 16);
        const int satd = (abs((int32_t)fdec_acs - (int32_t)fenc_acs) + abs((int32_t)(fdec_acs >> 32) - (int32_t)(fenc_acs >> 32))) >> 1;
        ssd += (satd * e->cfg.psy_rd_q8 * a->lambda + 128) >> 8;
    }
    int cssd = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 16; x++) { const int d = fuv[y * e->fs + x] - ruv[y * e->rs + x]; cssd += d * d; }
    return ssd + (int)(((int64_t)cssd * a->chroma_lambda2_offset + 128) >> 8);
}

static void cabac_ctx_of(const actx *a, x264o_cabac_ctx *cc, uint8_t *state)
{
    x264o_encoder *e = a->e;
    cc->mbs = e->mbs; cc->levels = e->levels; cc->mbw = e->mbw; cc->mbh = e->mbh; cc->first_row = e->row0;
    cc->pslice = e->slice_type == X264GPU_SLICE_P; cc->num_ref = e->nref; cc->t8mode = e->cfg.dct8x8;
    cc->amvd = e->amvd; cc->state = state; cc->last_dqp = e->last_dqp; cc->last_qp = e->last_qp;
    cc->bslice = e->slice_type == X264GPU_SLICE_B; cc->num_ref1 = e->nref_l[1]; cc->amvd1 = e->amvd1;
    if (cc->bslice) cc->pslice = 1;      /* an inter slice: the P / B context initialisation */
}

static void rd_reset(const actx *a, x264gpu_mb *mb, int16_t *lv)
{
    memset(mb, 0, sizeof(*mb));
    mb->qp = (uint8_t)a->qp;
    memset(lv, 0, X264GPU_MB_LEVELS * sizeof(int16_t));
}

/* what a whole-macroblock encode leaves in x264's non_zero_count cache (CABAC: flags; an 8x8 transform block sets its four entries alike) */
static void nnzc_sync(actx *a, const x264gpu_mb *mb)
{
    const int skip = mb->type == X264GPU_MB_P_SKIP || mb->type == X264GPU_MB_B_SKIP;
    for (int b = 0; b < 16; b++) a->nnzc[b] = (uint8_t)(!skip && ((mb->cbp_luma >> (b >> 2)) & 1) ? (mb->transform8x8 ? 1 : (mb->nnz >> b) & 1) : 0);
    for (int i = 0; i < 8; i++) a->nnzc[16 + i] = (uint8_t)(!skip && mb->cbp_chroma == 2 ? (mb->nnz >> (16 + i)) & 1 : 0);
}

/* x264_macroblock_deblock ([x264-upstream] encoder/macroblock.c; h->mb.b_deblock_rdo: --subme 9 and up, cfg.rd bit 6): before a whole-macroblock RD
 * candidate's distortion is measured its luma is loop-filtered along the INTERNAL edges (what the real filter will do to them later; the macroblock's
 * outer edges need the neighbours' final state and are left out): boundary strength 3 everywhere for intra, else from the coded flags / reference
 * indices / vector differences of the 4x4 blocks on either side (deblock_strength: 2, 1 or 0; reference INDICES as they are, both lists in B slices);
 * edges 1 and 3 only under the 4x4 transform; nothing for an uncoded 16x16 or at quantisers the filter does not touch */
static void macroblock_deblock(actx *a, const x264gpu_mb *mb)
{
    x264o_encoder *e = a->e;
    const int aoff = e->cfg.deblock_alpha * 2, boff = e->cfg.deblock_beta * 2;
    const int qp_thresh = 15 - (aoff < boff ? aoff : boff) - (e->cfg.chroma_qp_offset > 0 ? e->cfg.chroma_qp_offset : 0);
    const int intra = is_intra_type(mb->type), qp = a->qp;
    if (!e->cfg.deblock) return;
    if ((mb->partition == D_16x16 && !mb->cbp_luma && !intra) || qp <= qp_thresh) return;
    const int ia = clampi(qp + aoff, 0, 51), ib = clampi(qp + boff, 0, 51);
    const int alpha = x264o_alpha_table[ia], beta = x264o_beta_table[ib];
    if (!alpha || !beta) return;
    const int t8 = mb->transform8x8, bframe = e->slice_type == X264GPU_SLICE_B;
    /* coded flag of the 4x4 block at (x, y): an 8x8-transform block flags all four (STORE_8x8_NNZ) */
    int nz[4][4];
    for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        const int i8 = (y >> 1) * 2 + (x >> 1), blk = i8 * 4 + (y & 1) * 2 + (x & 1);
        nz[y][x] = t8 ? (mb->cbp_luma >> i8) & 1 : (mb->nnz >> blk) & 1;
    }
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    for (int dir = 0; dir < 2; dir++)
        for (int edge = 1; edge < 4; edge++) {
            if (t8 && (edge & 1)) continue;
            for (int i = 0; i < 4; i++) {
                const int x = dir ? i : edge, y = dir ? edge : i, xn = dir ? x : x - 1, yn = dir ? y - 1 : y;
                int bs;
                if (intra) bs = 3;
                else if (nz[y][x] || nz[yn][xn]) bs = 2;
                else {
                    const int k = (y >> 1) * 2 + (x >> 1), kn = (yn >> 1) * 2 + (xn >> 1);
                    bs = mb->ref[k] != mb->ref[kn] || abs(mb->mv[k][0] - mb->mv[kn][0]) >= 4 || abs(mb->mv[k][1] - mb->mv[kn][1]) >= 4 ||
                         (bframe && (mb->ref1[k] != mb->ref1[kn] || abs(mb->mv1[k][0] - mb->mv1[kn][0]) >= 4 || abs(mb->mv1[k][1] - mb->mv1[kn][1]) >= 4));
                }
                if (!bs) continue;
                pixel *q0 = rec + (size_t)(4 * y) * e->rs + 4 * x;
                x264o_deblock_luma_edge(q0, dir ? e->rs : 1, dir ? 1 : e->rs, 4, alpha, beta, x264o_tc0_table[ia][bs - 1], bs);
            }
        }
}

static int rd_finish(actx *a, x264gpu_mb *mb, int16_t *lv)
{
    if (a->e->cfg.rd & 64) macroblock_deblock(a, mb);
    const int ssd = rd_ssd_mb(a);
    nnzc_sync(a, mb);
    if (mb->type == X264GPU_MB_P_SKIP || mb->type == X264GPU_MB_B_SKIP) return ssd + ((a->lambda2 + 128) >> 8);
    if (a->e->cfg.cabac) {
        /* x264_rd_cost_mb under CABAC: the macroblock's syntax priced on a copy of the slice's context states, 1/256 bit units */
        uint8_t st[460];
        x264o_cabac_ctx cc;
        memcpy(st, a->e->cabac_state, sizeof(st));
        cabac_ctx_of(a, &cc, st);
        return ssd + (int)(((uint64_t)x264o_cabac_mb(&cc, a->mbx, a->mby, 1) * (uint64_t)a->lambda2 + 32768) >> 16);
    }
    return ssd + (int)(((int64_t)mb_bits_cavlc(a, mb, lv) * a->lambda2 + 128) >> 8);
}

/* x264_rd_cost_mb of an inter candidate: partition + the vectors of its search results, transform size t8 */
static int rd_cost_inter(actx *a, int partition, int t8, x264gpu_mb *mb, int16_t *lv)
{
    rd_reset(a, mb, lv);
    mb->type = partition == D_8x8 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0; mb->partition = (uint8_t)partition;
    for (int k = 0; k < 4; k++) {
        const me_t *m = partition == D_16x16 ? &a->me16 : partition == D_16x8 ? &a->me16x8[k >> 1] : partition == D_8x16 ? &a->me8x16[k & 1] : &a->me8[k];
        mb->ref[k] = (int8_t)m->ref; mb->mv[k][0] = (int16_t)m->mv[0]; mb->mv[k][1] = (int16_t)m->mv[1];
    }
    a->force_t8 = t8;
    encode_inter_mb(a, mb, lv);
    a->force_t8 = -1;
    return rd_finish(a, mb, lv);
}

static int rd_cost_intra(actx *a, int type, x264gpu_mb *mb, int16_t *lv)
{
    rd_reset(a, mb, lv);
    encode_intra_mb(a, type, mb, lv);
    return rd_finish(a, mb, lv);
}

/* x264_intra_rd: the SATD costs of the intra types within reach become RD costs, the others drop out */
static void intra_rd(actx *a, int thresh, x264gpu_mb *mb, int16_t *lv)
{
    if (!a->b_early_terminate) thresh = COST_MAX;
    a->satd_i16 = a->satd_i16 < thresh ? rd_cost_intra(a, X264GPU_MB_I16x16, mb, lv) : COST_MAX;
    a->satd_i4 = a->satd_i4 < thresh ? rd_cost_intra(a, X264GPU_MB_I4x4, mb, lv) : COST_MAX;
    a->satd_i8 = a->satd_i8 < thresh ? rd_cost_intra(a, X264GPU_MB_I8x8, mb, lv) : COST_MAX;
    if (a->satd_i8 < COST_MAX) a->cbp_i8 = mb->cbp_luma;          /* a->i_cbp_i8x8_luma */
}

/* x264_mb_analyse_p_rd */
static void analyse_p_rd(actx *a, int i_satd, x264gpu_mb *mb, int16_t *lv)
{
    const int thresh = a->b_early_terminate ? i_satd * 5 / 4 + 1 : COST_MAX;
    if (a->rd16x16 == COST_MAX && (!a->b_early_terminate || a->me16.cost <= i_satd * 3 / 2)) a->rd16x16 = rd_cost_inter(a, D_16x16, 0, mb, lv);
    a->cost16x8 = a->cost16x8 < thresh ? rd_cost_inter(a, D_16x8, 0, mb, lv) : COST_MAX;
    a->cost8x16 = a->cost8x16 < thresh ? rd_cost_inter(a, D_8x16, 0, mb, lv) : COST_MAX;
    a->cost8x8 = a->cost8x8 < thresh ? rd_cost_inter(a, D_8x8, 0, mb, lv) : COST_MAX;
}

/* ===========================================================================================================================
 * B slices ([x264-upstream] encoder/analyse.c x264_macroblock_analyse, SLICE_TYPE_B branch with i_mbrd 1: x264_mb_predict_mv_direct16x16
 * (spatial), x264_mb_analyse_inter_direct / _b16x16 / _b8x8_mixed_ref / _b8x8 / _b16x8 / _b8x16, x264_mb_analyse_b_rd, x264_refine_bidir;
 * encoder/me.c x264_me_refine_bidir_satd).  Restated from memory (oracle/BFRAME_NOTES.md): parity unpinned.  RD sessions only. */
/* B slices: --partitions b8x8 (x264 X264_ANALYSE_BSUB16x16) is bit 11 of cfg.partitions when bit 8 marks the extended set; else as p8x8 */
static int b_sub16x16(const x264o_encoder *e) { return (e->cfg.partitions & 0x100) ? (e->cfg.partitions >> 11) & 1 : e->cfg.partitions & 1; }
static const uint8_t mb_b_cost_direct = 1, mb_b_cost_l0 = 3, mb_b_cost_l1 = 3, mb_b_cost_bi = 5, mb_b_cost_8x8 = 9;      /* i_mb_b_cost_table */
static const uint8_t sub_b_cost[4] = { 3, 3, 5, 1 };                          /* i_sub_mb_b_cost_table: L0_8x8, L1_8x8, BI_8x8, DIRECT_8x8 */
/* i_mb_b16x8_cost_table[B_L0_L0 + 3 * first + second]: first / second half from list 0, list 1, both */
static const uint8_t mb_b16x8_cost[9] = { 5, 7, 7, 7, 5, 7, 9, 9, 9 };

static void cache_block_l(actx *a, int l, int bx8, int by8, int w8, int h8, int ref, const int mv[2])
{
    for (int y = by8; y < by8 + h8; y++)
        for (int x = bx8; x < bx8 + w8; x++) {
            nb_t *n = l ? &a->cur8b[y * 2 + x] : &a->cur8[y * 2 + x];
            n->ref = ref;
            if (mv) { n->mv[0] = mv[0]; n->mv[1] = mv[1]; } else if (ref < 0) n->mv[0] = n->mv[1] = 0;
            if (l) a->cur_valid1 |= 1 << (y * 2 + x); else a->cur_valid |= 1 << (y * 2 + x);
        }
}

/* x264_mb_predict_mv_direct16x16, spatial mode (8.4.1.2.2): a->direct_ref[], a->direct_mv[][8x8 block][] */
static void predict_direct_spatial(actx *a)
{
    const x264o_encoder *e = a->e;
    const int gx = 2 * a->mbx, gy = 2 * a->mby;
    int ref[2], mv[2][2];
    for (int l = 0; l < 2; l++) {
        nb_t A = nb8l(a, l, gx - 1, gy), B = nb8l(a, l, gx, gy - 1), C = nb8l(a, l, gx + 2, gy - 1);
        if (C.ref == -2) C = nb8l(a, l, gx - 1, gy - 1);
        unsigned r = (unsigned)A.ref < (unsigned)B.ref ? (unsigned)A.ref : (unsigned)B.ref;      /* the smallest index in use: negatives are the largest */
        if ((unsigned)C.ref < r) r = (unsigned)C.ref;
        int i_ref = (int)r;
        mv[l][0] = mv[l][1] = 0;
        if (i_ref < 0) i_ref = -1;
        else {
            const int cnt = (A.ref == i_ref) + (B.ref == i_ref) + (C.ref == i_ref);
            if (cnt > 1) median_mv(mv[l], &A, &B, &C);
            else { const nb_t *s = A.ref == i_ref ? &A : B.ref == i_ref ? &B : &C; mv[l][0] = s->mv[0]; mv[l][1] = s->mv[1]; }
        }
        ref[l] = i_ref;
    }
    if (ref[0] < 0 && ref[1] < 0) ref[0] = ref[1] = 0;       /* nothing around: both lists, index 0, zero vectors */
    for (int l = 0; l < 2; l++) for (int k = 0; k < 4; k++) { a->direct_ref[l][k] = ref[l]; a->direct_mv[l][k][0] = mv[l][0]; a->direct_mv[l][k][1] = mv[l][1]; }
    if (!(mv[0][0] | mv[0][1] | mv[1][0] | mv[1][1]) || (ref[0] && ref[1])) return;
    /* colZeroFlag per 8x8 block (direct_8x8_inference: its corner): the co-located block of list 1's first picture points into ITS reference 0
     * with a vector within +-1 -> the vectors of the lists whose direct reference is 0 become zero */
    const int cs = ref_slot_l(e, 1, 0);
    for (int k = 0; k < 4; k++) {
        if (e->colref[cs][a->mi][k] != 0) continue;
        if (abs(e->colmv[cs][a->mi][k][0]) > 1 || abs(e->colmv[cs][a->mi][k][1]) > 1) continue;
        for (int l = 0; l < 2; l++) if (ref[l] == 0) a->direct_mv[l][k][0] = a->direct_mv[l][k][1] = 0;
    }
}

/* x264_mb_predict_mv_direct16x16, temporal mode (mb_predict_mv_direct16x16_temporal; 8.4.1.2.3 with direct_8x8_inference): per 8x8 block the
 * co-located block of list 1's first picture — its list-0 reference mapped into this picture's list 0, its vector scaled by the POC distances for
 * list 0, the remainder for list 1 (index 0).  An intra co-located macroblock gives index 0 and zero vectors.  Returns 0 (no direct prediction for
 * this macroblock) when a co-located block has no list-0 motion or its reference is not in this picture's list 0, as x264 does. */
static int predict_direct_temporal(actx *a)
{
    const x264o_encoder *e = a->e;
    const int cs = ref_slot_l(e, 1, 0);
    const int intra_col = is_intra_type(e->mbtype[cs][a->mi]);
    for (int k = 0; k < 4; k++) {
        a->direct_ref[1][k] = 0;
        if (intra_col) { a->direct_ref[0][k] = 0; a->direct_mv[0][k][0] = a->direct_mv[0][k][1] = a->direct_mv[1][k][0] = a->direct_mv[1][k][1] = 0; continue; }
        const int rc = e->colref0[cs][a->mi][k];
        const int i_ref = rc < 0 ? -1 : e->map_col_to_list0[rc];
        if (i_ref < 0) return 0;
        const int dsf = e->dist_scale[i_ref], cx = e->colmv[cs][a->mi][k][0], cy = e->colmv[cs][a->mi][k][1];
        const int l0x = (dsf * cx + 128) >> 8, l0y = (dsf * cy + 128) >> 8;
        a->direct_ref[0][k] = i_ref;
        a->direct_mv[0][k][0] = (int16_t)l0x; a->direct_mv[0][k][1] = (int16_t)l0y;
        a->direct_mv[1][k][0] = (int16_t)(l0x - cx); a->direct_mv[1][k][1] = (int16_t)(l0y - cy);
    }
    return 1;
}

static void set_direct_record(const actx *a, x264gpu_mb *mb, int k)
{
    mb->ref[k] = (int8_t)a->direct_ref[0][k]; mb->ref1[k] = (int8_t)a->direct_ref[1][k];
    mb->mv[k][0] = (int16_t)(a->direct_ref[0][k] < 0 ? 0 : a->direct_mv[0][k][0]); mb->mv[k][1] = (int16_t)(a->direct_ref[0][k] < 0 ? 0 : a->direct_mv[0][k][1]);
    mb->mv1[k][0] = (int16_t)(a->direct_ref[1][k] < 0 ? 0 : a->direct_mv[1][k][0]); mb->mv1[k][1] = (int16_t)(a->direct_ref[1][k] < 0 ? 0 : a->direct_mv[1][k][1]);
}

/* the record of a B candidate: type / partition / per-block list use from the analysis state (x264_analyse_update_cache) */
static void fill_b_record(const actx *a, int type, int partition, x264gpu_mb *mb)
{
    mb->type = (uint8_t)type; mb->partition = (uint8_t)partition; mb->direct8 = 0;
    for (int k = 0; k < 4; k++) {
        int use;      /* 0 list 0, 1 list 1, 2 both, 3 direct */
        const me_t *m0, *m1;
        if (type == X264GPU_MB_B_DIRECT || type == X264GPU_MB_B_SKIP) use = 3;
        else if (partition == D_16x16) use = a->sub8[0];      /* fill_b16: sub8[0] carries L0 / L1 / BI of the 16x16 candidate */
        else if (partition == D_16x8) use = a->part16x8[k >> 1];
        else if (partition == D_8x16) use = a->part8x16[k & 1];
        else use = a->sub8[k];
        if (use == 3) { set_direct_record(a, mb, k); mb->direct8 |= (uint8_t)(1 << k); continue; }
        if (partition == D_16x16) { m0 = use == 2 ? &a->bi16[0] : &a->me16l[0]; m1 = use == 2 ? &a->bi16[1] : &a->me16l[1]; }
        else if (partition == D_16x8) { m0 = &a->me16x8l[0][k >> 1]; m1 = &a->me16x8l[1][k >> 1]; }
        else if (partition == D_8x16) { m0 = &a->me8x16l[0][k & 1]; m1 = &a->me8x16l[1][k & 1]; }
        else { m0 = &a->me8l[0][k]; m1 = &a->me8l[1][k]; }
        mb->ref[k] = -1; mb->ref1[k] = -1; mb->mv[k][0] = mb->mv[k][1] = 0; mb->mv1[k][0] = mb->mv1[k][1] = 0;
        if (use != 1) { mb->ref[k] = (int8_t)m0->ref; mb->mv[k][0] = (int16_t)m0->mv[0]; mb->mv[k][1] = (int16_t)m0->mv[1]; }
        if (use != 0) { mb->ref1[k] = (int8_t)m1->ref; mb->mv1[k][0] = (int16_t)m1->mv[0]; mb->mv1[k][1] = (int16_t)m1->mv[1]; }
    }
}

/* x264_rd_cost_mb of a B candidate (transform size t8) */
static int rd_cost_b(actx *a, int type, int partition, int use16, int t8, x264gpu_mb *mb, int16_t *lv)
{
    rd_reset(a, mb, lv);
    const int bak = a->sub8[0];
    if (partition == D_16x16 && type == X264GPU_MB_B_INTER) a->sub8[0] = use16;
    fill_b_record(a, type, partition, mb);
    a->sub8[0] = bak;
    a->force_t8 = t8;
    encode_inter_mb(a, mb, lv);
    a->force_t8 = -1;
    return rd_finish(a, mb, lv);
}

/* the average of two motion-compensated blocks (get_ref x 2 + mc.avg) */
static void bi_pred(const actx *a, const me_t *m0, const me_t *m1, pixel *dst /* stride 16 */)
{
    const x264o_encoder *e = a->e;
    pixel p0[256], p1[256];
    sctx S;
    sctx_init(&S, a, (me_t *)m0); get_ref(&S, p0, m0->mv[0], m0->mv[1]);
    sctx_init(&S, a, (me_t *)m1); get_ref(&S, p1, m1->mv[0], m1->mv[1]);
    x264o_pixel_avg_weight(dst, 16, p0, 16, p1, 16, m0->w, m0->h, e->bipred_weight[m0->ref][m1->ref]);
}
/* analyse_bi_chroma: the chroma share of a bi-predicted block's SATD cost */
static int bi_chroma(const actx *a, const me_t *m0, const me_t *m1)
{
    const x264o_encoder *e = a->e;
    pixel u0[64], v0[64], u1[64], v1[64], bu[64], bv[64], fu[64], fv[64];
    const int cw = m0->w / 2, ch = m0->h / 2;
    const pixel *fuv = e->fenc_uv + (size_t)(a->mby * 8 + m0->oy / 2) * e->fs + a->mbx * 16 + m0->ox;
    for (int y = 0; y < ch; y++) for (int x = 0; x < cw; x++) { fu[y * 8 + x] = fuv[y * e->fs + 2 * x]; fv[y * 8 + x] = fuv[y * e->fs + 2 * x + 1]; }
    x264o_mc_chroma(u0, v0, 8, chroma_plane(e, ref_slot_l(e, 0, m0->ref)), e->rs, a->mbx * 8 + m0->ox / 2, a->mby * 8 + m0->oy / 2, m0->mv[0], m0->mv[1], cw, ch);
    x264o_mc_chroma(u1, v1, 8, chroma_plane(e, ref_slot_l(e, 1, m1->ref)), e->rs, a->mbx * 8 + m1->ox / 2, a->mby * 8 + m1->oy / 2, m1->mv[0], m1->mv[1], cw, ch);
    x264o_pixel_avg_weight(bu, 8, u0, 8, u1, 8, cw, ch, e->bipred_weight[m0->ref][m1->ref]);
    x264o_pixel_avg_weight(bv, 8, v0, 8, v1, 8, cw, ch, e->bipred_weight[m0->ref][m1->ref]);
    return mbcmp(a, fu, 8, bu, 8, cw, ch) + mbcmp(a, fv, 8, bv, 8, cw, ch);
}

/* x264_mb_analyse_inter_direct: the direct prediction is in the reconstruction buffers */
static void analyse_inter_direct(actx *a)
{
    const x264o_encoder *e = a->e;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16, *fuv = e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16;
    const pixel *rec = luma_plane((x264o_encoder *)e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    const pixel *ruv = chroma_plane((x264o_encoder *)e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    a->cost16x16direct = a->lambda * mb_b_cost_direct;
    if (b_sub16x16(e)) {
        for (int i = 0; i < 4; i++) {
            const int x = (i & 1) * 8, y = (i >> 1) * 8;
            a->cost8x8direct[i] = mbcmp(a, fenc + y * e->fs + x, e->fs, rec + y * e->rs + x, e->rs, 8, 8);
            if (a->chroma_me) {
                pixel fu[16], fv[16], pu[16], pv[16];
                for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
                    fu[r * 4 + c] = fuv[(y / 2 + r) * e->fs + 2 * (x / 2 + c)]; fv[r * 4 + c] = fuv[(y / 2 + r) * e->fs + 2 * (x / 2 + c) + 1];
                    pu[r * 4 + c] = ruv[(y / 2 + r) * e->rs + 2 * (x / 2 + c)]; pv[r * 4 + c] = ruv[(y / 2 + r) * e->rs + 2 * (x / 2 + c) + 1];
                }
                a->cost8x8direct[i] += mbcmp(a, fu, 4, pu, 4, 4, 4) + mbcmp(a, fv, 4, pv, 4, 4, 4);
            }
            a->cost16x16direct += a->cost8x8direct[i];
            a->cost8x8direct[i] += a->lambda * sub_b_cost[3];
        }
    } else {
        a->cost16x16direct += mbcmp(a, fenc, e->fs, rec, e->rs, 16, 16);
        if (a->chroma_me) {
            pixel fu[64], fv[64], pu[64], pv[64];
            for (int r = 0; r < 8; r++) for (int c = 0; c < 8; c++) {
                fu[r * 8 + c] = fuv[r * e->fs + 2 * c]; fv[r * 8 + c] = fuv[r * e->fs + 2 * c + 1];
                pu[r * 8 + c] = ruv[r * e->rs + 2 * c]; pv[r * 8 + c] = ruv[r * e->rs + 2 * c + 1];
            }
            a->cost16x16direct += mbcmp(a, fu, 8, pu, 8, 8, 8) + mbcmp(a, fv, 8, pv, 8, 8, 8);
        }
    }
}

/* x264_mb_analyse_inter_b16x16.  The lists are searched list 1 first; with try_skip (analysis without RD, subme >= 3, the direct prediction
 * probed clean) the order is list 1 reference 0, list 0 reference 0, "both within 1 of the direct vectors -> B_SKIP" (returns 1), the rest of
 * list 0, the rest of list 1 */
static int analyse_inter_b16x16(actx *a, int try_skip)
{
    x264o_encoder *e = a->e;
    int mvc[9][2];
    int list1_skipped = 0;
    int i_halfpel_thresh[2] = { 0x7fffffff, 0x7fffffff };
    a->partition = D_16x16;
    a->me16l[0].cost = a->me16l[1].cost = 0x7fffffff;
    for (int l = 1; l >= 0;) {
        int *p_halfpel_thresh = (a->b_early_terminate && a->nref_l[l] > 1) ? &i_halfpel_thresh[l] : NULL;
        int r;
        for (r = (list1_skipped && l == 1) ? 1 : 0; r < a->nref_l[l]; r++) {
            if (try_skip && l == 1 && r > 0) { list1_skipped = 1; break; }
            me_t m;
            memset(&m, 0, sizeof(m));
            m.w = m.h = 16; m.list = l; m.ref = r; m.ref_cost = ref_cost_l(a, l, r);
            a->cur_valid = a->cur_valid1 = 0;
            predict_mv_l(a, l, 0, 0, 2, r, m.mvp);
            const int i_mvc = predict_mv_ref16x16_l(a, l, r, mvc);
            me_search_ref(a, &m, mvc, i_mvc, p_halfpel_thresh);
            m.cost += m.ref_cost;
            if (m.cost < a->me16l[l].cost) a->me16l[l] = m;
            a->mvcl[l][r][0][0] = m.mv[0]; a->mvcl[l][r][0][1] = m.mv[1];
            int16_t (*mvr)[2] = mvr_of(e, l, r);
            mvr[a->mi][0] = (int16_t)m.mv[0]; mvr[a->mi][1] = (int16_t)m.mv[1];
            if (r == 0 && try_skip) {          /* fast skip detection against the direct vector of the first 8x8 block */
                const int dx = a->direct_ref[l][0] < 0 ? 0 : a->direct_mv[l][0][0], dy = a->direct_ref[l][0] < 0 ? 0 : a->direct_mv[l][0][1];
                if (abs(a->me16l[l].mv[0] - dx) + abs(a->me16l[l].mv[1] - dy) > 1) try_skip = 0;
                else if (!l) return 1;         /* (the skip itself was tested before) */
            }
        }
        if (list1_skipped && l == 1 && r == a->nref_l[1]) break;
        if (list1_skipped && l == 0) l = 1; else l--;
    }
    /* the bi-predictive 16x16: both lists' winners averaged */
    a->bi16[0] = a->me16l[0]; a->bi16[1] = a->me16l[1];
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16;
    const int ref_costs = ref_cost_l(a, 0, a->bi16[0].ref) + ref_cost_l(a, 1, a->bi16[1].ref);
    pixel pix[256];
    bi_pred(a, &a->bi16[0], &a->bi16[1], pix);
    a->cost16x16bi = mbcmp(a, fenc, e->fs, pix, 16, 16, 16) + ref_costs + a->bi16[0].cost_mv + a->bi16[1].cost_mv;
    if (a->chroma_me) a->cost16x16bi += bi_chroma(a, &a->bi16[0], &a->bi16[1]);
    /* always try the zero vectors */
    if (a->bi16[0].mv[0] | a->bi16[0].mv[1] | a->bi16[1].mv[0] | a->bi16[1].mv[1]) {
        const int l0_mv_cost = a->cost_mv[-a->bi16[0].mvp[0]] + a->cost_mv[-a->bi16[0].mvp[1]];
        const int l1_mv_cost = a->cost_mv[-a->bi16[1].mvp[0]] + a->cost_mv[-a->bi16[1].mvp[1]];
        me_t z0 = a->bi16[0], z1 = a->bi16[1];
        z0.mv[0] = z0.mv[1] = z1.mv[0] = z1.mv[1] = 0;
        bi_pred(a, &z0, &z1, pix);
        int cost00 = mbcmp(a, fenc, e->fs, pix, 16, 16, 16) + ref_costs + l0_mv_cost + l1_mv_cost;
        if (a->chroma_me && cost00 < a->cost16x16bi) cost00 += bi_chroma(a, &z0, &z1);
        if (cost00 < a->cost16x16bi) {
            a->bi16[0].mv[0] = a->bi16[0].mv[1] = a->bi16[1].mv[0] = a->bi16[1].mv[1] = 0;
            a->bi16[0].cost_mv = l0_mv_cost; a->bi16[1].cost_mv = l1_mv_cost;
            a->cost16x16bi = cost00;
        }
    }
    a->cost16x16bi += a->lambda * mb_b_cost_bi;
    a->me16l[0].cost += a->lambda * mb_b_cost_l0;
    a->me16l[1].cost += a->lambda * mb_b_cost_l1;
    return 0;
}

/* mb_cache_mv_b8x8 / _b16x8 / _b8x16 without the mvd side: the block's part of both lists' motion cache */
static void cache_b_block(actx *a, int bx8, int by8, int w8, int h8, int use, const me_t *m0, const me_t *m1, int k_direct)
{
    if (use == 3) {
        for (int l = 0; l < 2; l++) cache_block_l(a, l, bx8, by8, w8, h8, a->direct_ref[l][k_direct], a->direct_ref[l][k_direct] < 0 ? NULL : a->direct_mv[l][k_direct]);
        return;
    }
    if (use != 1) cache_block_l(a, 0, bx8, by8, w8, h8, m0->ref, m0->mv); else cache_block_l(a, 0, bx8, by8, w8, h8, -1, NULL);
    if (use != 0) cache_block_l(a, 1, bx8, by8, w8, h8, m1->ref, m1->mv); else cache_block_l(a, 1, bx8, by8, w8, h8, -1, NULL);
}

/* x264_mb_analyse_inter_b8x8_mixed_ref / x264_mb_analyse_inter_b8x8 */
static void analyse_inter_b8x8(actx *a)
{
    x264o_encoder *e = a->e;
    const int mixed = e->cfg.mixed_refs;
    int i_maxref[2] = { a->nref_l[0] - 1, a->nref_l[1] - 1 };
    if (mixed)
        for (int l = 0; l < 2; l++)
            if (i_maxref[l] > 0 && a->me16l[l].ref == 0 && a->type_top > 0 && a->type_left > 0) {
                const int gx = 2 * a->mbx, gy = 2 * a->mby;
                const nb_t n[6] = { nb8l(a, l, gx - 1, gy - 1), nb8l(a, l, gx, gy - 1), nb8l(a, l, gx + 1, gy - 1), nb8l(a, l, gx + 2, gy - 1), nb8l(a, l, gx - 1, gy), nb8l(a, l, gx - 1, gy + 1) };
                i_maxref[l] = 0;
                for (int i = 0; i < 6; i++) if (n[i].ref > i_maxref[l]) i_maxref[l] = n[i].ref;
            }
    a->partition = D_8x8;
    a->cost8x8bi = 0;
    a->cur_valid = a->cur_valid1 = 0;
    for (int i = 0; i < 4; i++) {
        const int x8 = i & 1, y8 = i >> 1;
        const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + 8 * y8) * e->fs + a->mbx * 16 + 8 * x8;
        for (int l = 0; l < 2; l++) {
            me_t *lm = &a->me8l[l][i];
            if (mixed) {
                lm->cost = 0x7fffffff;
                for (int r = 0; r <= i_maxref[l]; r++) {
                    me_t m;
                    memset(&m, 0, sizeof(m));
                    m.w = m.h = 8; m.ox = 8 * x8; m.oy = 8 * y8; m.list = l; m.ref = r; m.ref_cost = ref_cost_l(a, l, r);
                    (l ? a->cur8b : a->cur8)[i].ref = r;
                    predict_mv_l(a, l, x8, y8, 1, r, m.mvp);
                    me_search_ref(a, &m, a->mvcl[l][r], i + 1, NULL);
                    m.cost += m.ref_cost;
                    if (m.cost < lm->cost) { *lm = m; a->satd8x8b[l][i] = m.cost - (m.cost_mv + m.ref_cost); }
                    a->mvcl[l][r][i + 1][0] = m.mv[0]; a->mvcl[l][r][i + 1][1] = m.mv[1];
                }
            } else {
                const int r = a->me16l[l].ref;
                int mvc1[1][2] = { { a->me16l[l].mv[0], a->me16l[l].mv[1] } };
                memset(lm, 0, sizeof(*lm));
                lm->w = lm->h = 8; lm->ox = 8 * x8; lm->oy = 8 * y8; lm->list = l; lm->ref = r; lm->ref_cost = ref_cost_l(a, l, r);
                (l ? a->cur8b : a->cur8)[i].ref = r;
                predict_mv_l(a, l, x8, y8, 1, r, lm->mvp);
                me_search_ref(a, lm, mvc1, 1, NULL);
                a->satd8x8b[l][i] = lm->cost - lm->cost_mv;
                lm->cost += lm->ref_cost;
                cache_block_l(a, l, x8, y8, 1, 1, r, lm->mv);
                a->mvcl[l][r][i + 1][0] = lm->mv[0]; a->mvcl[l][r][i + 1][1] = lm->mv[1];
            }
        }
        /* both lists */
        pixel pix[256];
        bi_pred(a, &a->me8l[0][i], &a->me8l[1][i], pix);
        a->satd8x8b[2][i] = mbcmp(a, fenc, e->fs, pix, 16, 8, 8);
        int i_part_cost_bi = a->satd8x8b[2][i] + a->me8l[0][i].cost_mv + a->me8l[1][i].cost_mv + a->me8l[0][i].ref_cost + a->me8l[1][i].ref_cost + a->lambda * sub_b_cost[2];
        if (a->chroma_me) { const int cc = bi_chroma(a, &a->me8l[0][i], &a->me8l[1][i]); i_part_cost_bi += cc; a->satd8x8b[2][i] += cc; }
        a->me8l[0][i].cost += a->lambda * sub_b_cost[0];
        a->me8l[1][i].cost += a->lambda * sub_b_cost[1];
        int i_part_cost = a->me8l[0][i].cost;
        a->sub8[i] = 0;
        if (a->me8l[1][i].cost < i_part_cost) { i_part_cost = a->me8l[1][i].cost; a->sub8[i] = 1; }
        if (i_part_cost_bi < i_part_cost) { i_part_cost = i_part_cost_bi; a->sub8[i] = 2; }
        if (a->cost8x8direct[i] < i_part_cost) { i_part_cost = a->cost8x8direct[i]; a->sub8[i] = 3; }
        a->cost8x8bi += i_part_cost;
        cache_b_block(a, x8, y8, 1, 1, a->sub8[i], &a->me8l[0][i], &a->me8l[1][i], i);
    }
    a->cost8x8bi += a->lambda * mb_b_cost_8x8;
}

/* x264_mb_analyse_inter_b16x8 (horizontal = 1) / _b8x16 (0) */
static void analyse_inter_b_halves(actx *a, int horizontal, int i_best_satd)
{
    x264o_encoder *e = a->e;
    int mvc[3][2];
    int *p_cost = horizontal ? &a->cost16x8bi : &a->cost8x16bi, *part = horizontal ? a->part16x8 : a->part8x16;
    const int *cost_est = horizontal ? a->cost_est16x8 : a->cost_est8x16;
    a->partition = horizontal ? D_16x8 : D_8x16;
    *p_cost = 0;
    a->cur_valid = a->cur_valid1 = 0;
    for (int i = 0; i < 2; i++) {
        const int bx8 = horizontal ? 0 : i, by8 = horizontal ? i : 0, w8 = horizontal ? 2 : 1, h8 = horizontal ? 1 : 2;
        const int ka = horizontal ? 2 * i : i, kb = horizontal ? 2 * i + 1 : i + 2;      /* the two 8x8 blocks the half covers */
        const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + 8 * by8) * e->fs + a->mbx * 16 + 8 * bx8;
        me_t *lm[2];
        for (int l = 0; l < 2; l++) {
            lm[l] = horizontal ? &a->me16x8l[l][i] : &a->me8x16l[l][i];
            const int ref8[2] = { a->me8l[l][ka].ref, a->me8l[l][kb].ref };
            const int i_ref8s = ref8[0] == ref8[1] ? 1 : 2;
            lm[l]->cost = 0x7fffffff;
            for (int j = 0; j < i_ref8s; j++) {
                const int r = ref8[j];
                me_t m;
                memset(&m, 0, sizeof(m));
                m.w = 8 * w8; m.h = 8 * h8; m.ox = 8 * bx8; m.oy = 8 * by8; m.list = l; m.ref = r; m.ref_cost = ref_cost_l(a, l, r);
                for (int k = 0; k < 2; k++) { mvc[0][k] = a->mvcl[l][r][0][k]; mvc[1][k] = a->mvcl[l][r][ka + 1][k]; mvc[2][k] = a->mvcl[l][r][kb + 1][k]; }
                (l ? a->cur8b : a->cur8)[ka].ref = (l ? a->cur8b : a->cur8)[kb].ref = r;
                predict_mv_l(a, l, bx8, by8, w8, r, m.mvp);
                me_search_ref(a, &m, mvc, 3, NULL);
                m.cost += m.ref_cost;
                if (m.cost < lm[l]->cost) *lm[l] = m;
            }
        }
        pixel pix[256];
        bi_pred(a, lm[0], lm[1], pix);
        int i_part_cost_bi = mbcmp(a, fenc, e->fs, pix, 16, 8 * w8, 8 * h8) + lm[0]->cost_mv + lm[1]->cost_mv + lm[0]->ref_cost + lm[1]->ref_cost;
        if (a->chroma_me) i_part_cost_bi += bi_chroma(a, lm[0], lm[1]);
        int i_part_cost = lm[0]->cost;
        part[i] = 0;
        if (lm[1]->cost < i_part_cost) { i_part_cost = lm[1]->cost; part[i] = 1; }
        if (i_part_cost_bi + a->lambda * 1 < i_part_cost) { i_part_cost = i_part_cost_bi; part[i] = 2; }
        *p_cost += i_part_cost;
        /* early termination: the first half plus the estimate of the second */
        if (a->b_early_terminate && !i && i_part_cost + cost_est[1] > i_best_satd * (16 + !!a->mbrd + (e->cfg.psy_rd_q8 != 0)) / 16) { *p_cost = COST_MAX; return; }
        cache_b_block(a, bx8, by8, w8, h8, part[i], lm[0], lm[1], 0);
    }
    *p_cost += a->lambda * mb_b16x8_cost[part[0] * 3 + part[1]];
}

/* x264_me_refine_bidir_satd: both vectors of a bi-predicted block walk together, up to two components at a time */
static void me_refine_bidir_satd(const actx *a, me_t *m0, me_t *m1, int i_weight)
{
    static const int8_t dia4d[33][4] = {
        { 0, 0, 0, 0 },
        { 0, 0, 0, 1 }, { 0, 0, 0, -1 }, { 0, 0, 1, 0 }, { 0, 0, -1, 0 }, { 0, 1, 0, 0 }, { 0, -1, 0, 0 }, { 1, 0, 0, 0 }, { -1, 0, 0, 0 },
        { 0, 0, 1, 1 }, { 0, 0, -1, -1 }, { 0, 1, 1, 0 }, { 0, -1, -1, 0 }, { 1, 1, 0, 0 }, { -1, -1, 0, 0 }, { 1, 0, 0, 1 }, { -1, 0, 0, -1 },
        { 0, 1, 0, 1 }, { 0, -1, 0, -1 }, { 1, 0, 1, 0 }, { -1, 0, -1, 0 }, { 0, 0, -1, 1 }, { 0, 0, 1, -1 }, { 0, -1, 1, 0 }, { 0, 1, -1, 0 },
        { -1, 1, 0, 0 }, { 1, -1, 0, 0 }, { 1, 0, 0, -1 }, { -1, 0, 0, 1 }, { 0, -1, 0, 1 }, { 0, 1, 0, -1 }, { -1, 0, 1, 0 }, { 1, 0, -1, 0 } };
    const x264o_encoder *e = a->e;
    const int bw = m0->w, bh = m0->h;
    int bm0x = m0->mv[0], bm0y = m0->mv[1], bm1x = m1->mv[0], bm1y = m1->mv[1], bcost = COST_MAX, mc_list0 = 1, mc_list1 = 1;
    if (bm0y < a->smin[1] + 8 || bm1y < a->smin[1] + 8 || bm0y > a->smax[1] - 8 || bm1y > a->smax[1] - 8 ||
        bm0x < a->smin[0] + 8 || bm1x < a->smin[0] + 8 || bm0x > a->smax[0] - 8 || bm1x > a->smax[0] - 8) return;
    const uint16_t *cm0x = a->cost_mv - m0->mvp[0], *cm0y = a->cost_mv - m0->mvp[1], *cm1x = a->cost_mv - m1->mvp[0], *cm1y = a->cost_mv - m1->mvp[1];
    const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + m0->oy) * e->fs + a->mbx * 16 + m0->ox;
    static pixel buf[2][9][256];
    uint8_t visited[8][8][8];
    memset(visited, 0, sizeof(visited));
    sctx S0, S1;
    sctx_init(&S0, a, m0); sctx_init(&S1, a, m1);
    for (int pass = 0; pass < 8; pass++) {
        int bestj = 0;
        /* the nine sub-pel neighbours of each list's vector (square1 order; index 4 + 3 dx + dy) */
        if (mc_list0) for (int j = 0; j < 9; j++) get_ref(&S0, buf[0][4 + 3 * square1[j][0] + square1[j][1]], bm0x + square1[j][0], bm0y + square1[j][1]);
        if (mc_list1) for (int j = 0; j < 9; j++) get_ref(&S1, buf[1][4 + 3 * square1[j][0] + square1[j][1]], bm1x + square1[j][0], bm1y + square1[j][1]);
        for (int j = !!pass; j < 33; j++) {
            const int m0x = dia4d[j][0] + bm0x, m0y = dia4d[j][1] + bm0y, m1x = dia4d[j][2] + bm1x, m1y = dia4d[j][3] + bm1y;
            if (!pass || !(visited[m0x & 7][m0y & 7][m1x & 7] & (1 << (m1y & 7)))) {
                const int i0 = 4 + 3 * dia4d[j][0] + dia4d[j][1], i1 = 4 + 3 * dia4d[j][2] + dia4d[j][3];
                pixel pix[256];
                visited[m0x & 7][m0y & 7][m1x & 7] |= (uint8_t)(1 << (m1y & 7));
                x264o_pixel_avg_weight(pix, 16, buf[0][i0], 16, buf[1][i1], 16, bw, bh, i_weight);
                const int cost = mbcmp(a, fenc, e->fs, pix, 16, bw, bh) + cm0x[m0x] + cm0y[m0y] + cm1x[m1x] + cm1y[m1y];
                if (cost < bcost) { bcost = cost; bestj = j; }
            }
        }
        if (!bestj) break;
        bm0x += dia4d[bestj][0]; bm0y += dia4d[bestj][1]; bm1x += dia4d[bestj][2]; bm1y += dia4d[bestj][3];
        mc_list0 = dia4d[bestj][0] | dia4d[bestj][1]; mc_list1 = dia4d[bestj][2] | dia4d[bestj][3];
    }
    m0->mv[0] = bm0x; m0->mv[1] = bm0y; m1->mv[0] = bm1x; m1->mv[1] = bm1y;
}

/* x264_mb_analyse_b_rd */
static void analyse_b_rd(actx *a, int i_satd_inter, x264gpu_mb *mb, int16_t *lv)
{
    const int thresh = a->b_early_terminate ? i_satd_inter * (17 + (a->e->cfg.psy_rd_q8 != 0)) / 16 + 1 : COST_MAX;
    if (a->b_direct_available && a->rd16direct == COST_MAX) a->rd16direct = rd_cost_b(a, X264GPU_MB_B_DIRECT, D_16x16, 0, 0, mb, lv);
    if (a->me16l[0].cost < thresh && a->rd16l[0] == COST_MAX) a->rd16l[0] = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x16, 0, 0, mb, lv);
    if (a->me16l[1].cost < thresh && a->rd16l[1] == COST_MAX) a->rd16l[1] = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x16, 1, 0, mb, lv);
    if (a->cost16x16bi < thresh && a->rd16bi == COST_MAX) a->rd16bi = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x16, 2, 0, mb, lv);
    if (a->cost8x8bi < thresh && a->rd8x8bi == COST_MAX) a->rd8x8bi = rd_cost_b(a, X264GPU_MB_B_8x8, D_8x8, 0, 0, mb, lv);
    if (a->cost16x8bi < thresh && a->rd16x8bi == COST_MAX) a->rd16x8bi = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x8, 0, 0, mb, lv);
    if (a->cost8x16bi < thresh && a->rd8x16bi == COST_MAX) a->rd8x16bi = rd_cost_b(a, X264GPU_MB_B_INTER, D_8x16, 0, 0, mb, lv);
}

static void intra_rd(actx *a, int thresh, x264gpu_mb *mb, int16_t *lv);
static void intra_rd_refine(actx *a, int type, x264gpu_mb *mb, int16_t *lv);
static void refine_inter_b_rd(actx *a, int i_type, int i_partition, int use16, int t8, int i_cost, x264gpu_mb *mb, int16_t *lv);

/* the B branch of x264_macroblock_analyse + x264_macroblock_encode; the caller has set up qp / lambda / limits */
static void macroblock_b(actx *a, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int mi = a->mi;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    pixel *rec_uv = chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
    a->nref_l[0] = e->nref_l[0]; a->nref_l[1] = e->nref_l[1];
    a->rd16l[0] = a->rd16l[1] = a->rd16bi = a->rd16direct = a->rd8x8bi = a->rd16x8bi = a->rd8x16bi = COST_MAX;
    a->cost8x8bi = a->cost16x8bi = a->cost8x16bi = COST_MAX;
    for (int k = 0; k < 4; k++) a->direct_ref[0][k] = a->direct_ref[1][k] = -1;
    /* direct prediction (x264_mb_predict_mv_direct16x16 in the slice's mode), motion-compensated into the reconstruction; B_SKIP when its distortion
     * alone is below the cheapest coded macroblock.  --direct auto (b_direct_auto_write): the other mode is predicted and probed first, then the
     * slice's own — "prefer whichever mode allows more Skip macroblocks" (h->stat.frame.i_direct_score) */
    a->cur_valid = a->cur_valid1 = 0;
    a->bskip_cost = COST_MAX; a->cost16x16direct = COST_MAX;
    for (int i = 0; i < 4; i++) a->cost8x8direct[i] = COST_MAX;
    int probed = 0;
    for (int pass = e->direct_auto ? 0 : 1; pass < 2; pass++) {
        const int temporal = pass ? e->direct_temporal : !e->direct_temporal;
        if (temporal) a->b_direct_available = predict_direct_temporal(a); else { predict_direct_spatial(a); a->b_direct_available = 1; }
        probed = 0;
        if (!a->b_direct_available) continue;
        x264gpu_mb t;
        pixel pu[64], pv[64];
        memset(&t, 0, sizeof(t));
        fill_b_record(a, X264GPU_MB_B_SKIP, D_16x16, &t);
        mc_mb_b(e, a->mbx, a->mby, &t, rec, e->rs, pu, pv);
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { rec_uv[y * e->rs + 2 * x] = pu[y * 8 + x]; rec_uv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
        if (e->direct_auto) { probed = probe_bskip(a); e->direct_score[!temporal] += probed; }       /* (score[1] = spatial) */
    }
    int b_skip = 0, try_skip = 0;
    if (a->b_direct_available) {
        if (a->mbrd) { a->bskip_cost = rd_ssd_mb(a); b_skip = a->bskip_cost <= ((6 * a->lambda2 + 128) >> 8); }
        else if (!e->direct_auto) {
            /* without RD: x264_macroblock_probe_bskip; from subme 3 on the skip also wants both 16x16 searches to land on the direct vectors */
            try_skip = probe_bskip(a);
            if (a->subme < 3) b_skip = try_skip;
        } else b_skip = probed;          /* (--direct auto without RD: the probe of the slice's mode stands) */
    }
    if (b_skip) {
        for (int l = 0; l < 2; l++) for (int r = 0; r < a->nref_l[l]; r++) { int16_t (*mvr)[2] = mvr_of(e, l, r); mvr[mi][0] = mvr[mi][1] = 0; }
        fill_b_record(a, X264GPU_MB_B_SKIP, D_16x16, mb);
        return;                       /* the prediction is the reconstruction */
    }
    if (a->b_direct_available)
    analyse_inter_direct(a);
    if (analyse_inter_b16x16(a, try_skip)) {
        for (int l = 0; l < 2; l++) for (int r = 1; r < a->nref_l[l]; r++) { int16_t (*mvr)[2] = mvr_of(e, l, r); mvr[mi][0] = mvr[mi][1] = 0; }
        fill_b_record(a, X264GPU_MB_B_SKIP, D_16x16, mb);
        return;                       /* (the direct prediction is still in the reconstruction) */
    }
    int use16 = 0, i_type = X264GPU_MB_B_INTER, i_partition = D_16x16, i_cost = a->me16l[0].cost;
    if (a->me16l[1].cost < i_cost) { i_cost = a->me16l[1].cost; use16 = 1; }
    if (a->cost16x16bi < i_cost) { i_cost = a->cost16x16bi; use16 = 2; }
    if (a->cost16x16direct < i_cost) { i_cost = a->cost16x16direct; i_type = X264GPU_MB_B_DIRECT; }
    if (a->mbrd && a->b_early_terminate && a->cost16x16direct <= i_cost * 33 / 32) {
        analyse_b_rd(a, i_cost, mb, lv);
        if (a->bskip_cost < a->rd16direct && a->bskip_cost < a->rd16bi && a->bskip_cost < a->rd16l[0] && a->bskip_cost < a->rd16l[1]) {
            rd_reset(a, mb, lv);
            fill_b_record(a, X264GPU_MB_B_SKIP, D_16x16, mb);
            encode_inter_mb(a, mb, lv);
            return;
        }
    }
    if (b_sub16x16(e)) {
        analyse_inter_b8x8(a);
        if (a->cost8x8bi < i_cost) { i_cost = a->cost8x8bi; i_type = X264GPU_MB_B_8x8; i_partition = D_8x8; }
        /* estimates of the two-partition shapes from the SATD scores of the 8x8 blocks: the likelier one is analysed first */
        int part_est16x8[2], part_est8x16[2];
        for (int i = 0; i < 2; i++) {
            for (int hor = 1; hor >= 0; hor--) {
                const int ka = hor ? 2 * i : i, kb = hor ? 2 * i + 1 : i + 2;
                const int l0_satd = a->satd8x8b[0][ka] + a->satd8x8b[0][kb], l1_satd = a->satd8x8b[1][ka] + a->satd8x8b[1][kb], bi_satd = a->satd8x8b[2][ka] + a->satd8x8b[2][kb];
                const int avg_l0 = (a->me8l[0][ka].cost_mv + a->me8l[0][ka].ref_cost + a->me8l[0][kb].cost_mv + a->me8l[0][kb].ref_cost + 1) >> 1;
                const int avg_l1 = (a->me8l[1][ka].cost_mv + a->me8l[1][ka].ref_cost + a->me8l[1][kb].cost_mv + a->me8l[1][kb].ref_cost + 1) >> 1;
                int best = COST_MAX, *pe = hor ? part_est16x8 : part_est8x16;
                if (l0_satd + avg_l0 < best) { best = l0_satd + avg_l0; pe[i] = 0; }
                if (l1_satd + avg_l1 < best) { best = l1_satd + avg_l1; pe[i] = 1; }
                if (bi_satd + avg_l0 + avg_l1 < best) { best = bi_satd + avg_l0 + avg_l1; pe[i] = 2; }
                (hor ? a->cost_est16x8 : a->cost_est8x16)[i] = best;
            }
        }
        a->cost_est16x8[1] += a->lambda * mb_b16x8_cost[part_est16x8[0] * 3 + part_est16x8[1]];
        a->cost_est8x16[1] += a->lambda * mb_b16x8_cost[part_est8x16[0] * 3 + part_est8x16[1]];
        const int est16x8 = a->cost_est16x8[0] + a->cost_est16x8[1], est8x16 = a->cost_est8x16[0] + a->cost_est8x16[1];
        const int try_16x8_first = est16x8 < est8x16;
        if (try_16x8_first && (!a->b_early_terminate || est16x8 < i_cost)) {
            analyse_inter_b_halves(a, 1, i_cost);
            if (a->cost16x8bi < i_cost) { i_cost = a->cost16x8bi; i_type = X264GPU_MB_B_INTER; i_partition = D_16x8; }
        }
        if (!a->b_early_terminate || est8x16 < i_cost) {
            analyse_inter_b_halves(a, 0, i_cost);
            if (a->cost8x16bi < i_cost) { i_cost = a->cost8x16bi; i_type = X264GPU_MB_B_INTER; i_partition = D_8x16; }
        }
        if (!try_16x8_first && (!a->b_early_terminate || est16x8 < i_cost)) {
            analyse_inter_b_halves(a, 1, i_cost);
            if (a->cost16x8bi < i_cost) { i_cost = a->cost16x8bi; i_type = X264GPU_MB_B_INTER; i_partition = D_16x8; }
        }
    }
    if (!a->mbrd) {
        /* ---- analysis without RD (x264 below subme 7 in B slices): quarter-pel refinement of the winner's vectors, intra on SATD cost ---- */
        if (a->subme) {
            if (i_partition == D_16x16 && i_type == X264GPU_MB_B_INTER) {
                a->partition = D_16x16;
                a->me16l[0].cost -= a->lambda * mb_b_cost_l0; a->me16l[1].cost -= a->lambda * mb_b_cost_l1;
                if (use16 == 0) { me_refine_qpel(a, &a->me16l[0]); i_cost = a->me16l[0].cost + a->lambda * mb_b_cost_l0; }
                else if (use16 == 1) { me_refine_qpel(a, &a->me16l[1]); i_cost = a->me16l[1].cost + a->lambda * mb_b_cost_l1; }
                else { me_refine_qpel(a, &a->bi16[0]); me_refine_qpel(a, &a->bi16[1]); }
            } else if (i_partition == D_16x8) {
                for (int i = 0; i < 2; i++) { if (a->part16x8[i] != 1) me_refine_qpel(a, &a->me16x8l[0][i]); if (a->part16x8[i] != 0) me_refine_qpel(a, &a->me16x8l[1][i]); }
            } else if (i_partition == D_8x16) {
                for (int i = 0; i < 2; i++) { if (a->part8x16[i] != 1) me_refine_qpel(a, &a->me8x16l[0][i]); if (a->part8x16[i] != 0) me_refine_qpel(a, &a->me8x16l[1][i]); }
            } else if (i_partition == D_8x8) {
                /* (x264 refines the unused vectors of a direct block as well; nothing reads them afterwards) */
                for (int i = 0; i < 4; i++) { if (a->sub8[i] == 0 || a->sub8[i] == 2) me_refine_qpel(a, &a->me8l[0][i]); if (a->sub8[i] == 1 || a->sub8[i] == 2) me_refine_qpel(a, &a->me8l[1][i]); }
            }
        }
        const int i_satd_inter0 = i_cost;
        if (a->chroma_me) {
            analyse_intra_chroma(a);
            analyse_intra(a, i_satd_inter0 - a->satd_chroma);
            a->satd_i16 += a->satd_chroma; a->satd_i8 += a->satd_chroma; a->satd_i4 += a->satd_chroma;
        } else analyse_intra(a, i_satd_inter0);
        int intra_type = -1;
        if (a->satd_i16 < i_cost) { i_cost = a->satd_i16; intra_type = X264GPU_MB_I16x16; }
        if (a->satd_i8 < i_cost) { i_cost = a->satd_i8; intra_type = X264GPU_MB_I8x8; }
        if (a->satd_i4 < i_cost) { i_cost = a->satd_i4; intra_type = X264GPU_MB_I4x4; }
        if (intra_type >= 0) {
            mb->cost = i_cost;
            e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
            encode_intra_mb(a, intra_type, mb, lv);
            e->b_trellis = 0;
            e->intra_count++;
            return;
        }
        if (a->subme >= 5 && i_type != X264GPU_MB_B_DIRECT) {
            if (i_partition == D_16x16) { if (use16 == 2) me_refine_bidir_satd(a, &a->bi16[0], &a->bi16[1], e->bipred_weight[a->bi16[0].ref][a->bi16[1].ref]); }
            else if (i_partition == D_16x8) { for (int i = 0; i < 2; i++) if (a->part16x8[i] == 2) me_refine_bidir_satd(a, &a->me16x8l[0][i], &a->me16x8l[1][i], e->bipred_weight[a->me16x8l[0][i].ref][a->me16x8l[1][i].ref]); }
            else if (i_partition == D_8x16) { for (int i = 0; i < 2; i++) if (a->part8x16[i] == 2) me_refine_bidir_satd(a, &a->me8x16l[0][i], &a->me8x16l[1][i], e->bipred_weight[a->me8x16l[0][i].ref][a->me8x16l[1][i].ref]); }
            else for (int i = 0; i < 4; i++) if (a->sub8[i] == 2) me_refine_bidir_satd(a, &a->me8l[0][i], &a->me8l[1][i], e->bipred_weight[a->me8l[0][i].ref][a->me8l[1][i].ref]);
        }
        {
            const int bak = a->sub8[0];
            if (i_partition == D_16x16 && i_type == X264GPU_MB_B_INTER) a->sub8[0] = use16;
            fill_b_record(a, i_type, i_partition, mb);
            a->sub8[0] = bak;
        }
        a->force_t8 = -1;              /* x264_mb_analyse_transform: SA8D against SATD of the prediction error */
        e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
        encode_inter_mb(a, mb, lv);
        e->b_trellis = 0;
        return;
    }
    int i_satd_inter = i_cost;
    /* RD: every candidate within reach, B_SKIP included */
    analyse_b_rd(a, i_satd_inter, mb, lv);
    i_type = X264GPU_MB_B_SKIP; i_cost = a->bskip_cost; i_partition = D_16x16;
    if (a->rd16l[0] < i_cost) { i_cost = a->rd16l[0]; i_type = X264GPU_MB_B_INTER; use16 = 0; }
    if (a->rd16l[1] < i_cost) { i_cost = a->rd16l[1]; i_type = X264GPU_MB_B_INTER; use16 = 1; }
    if (a->rd16bi < i_cost) { i_cost = a->rd16bi; i_type = X264GPU_MB_B_INTER; use16 = 2; }
    if (a->rd16direct < i_cost) { i_cost = a->rd16direct; i_type = X264GPU_MB_B_DIRECT; }
    if (a->rd16x8bi < i_cost) { i_cost = a->rd16x8bi; i_type = X264GPU_MB_B_INTER; i_partition = D_16x8; }
    if (a->rd8x16bi < i_cost) { i_cost = a->rd8x16bi; i_type = X264GPU_MB_B_INTER; i_partition = D_8x16; }
    if (a->rd8x8bi < i_cost) { i_cost = a->rd8x8bi; i_type = X264GPU_MB_B_8x8; i_partition = D_8x8; }
    /* intra analysis against the inter SATD cost */
    if (a->chroma_me) {
        analyse_intra_chroma(a);
        analyse_intra(a, i_satd_inter - a->satd_chroma);
        a->satd_i16 += a->satd_chroma; a->satd_i8 += a->satd_chroma; a->satd_i4 += a->satd_chroma;
    } else analyse_intra(a, i_satd_inter);
    /* x264_mb_analyse_transform_rd: the other transform size for the winner (B_SKIP has no transform) */
    int t8 = 0;
    if (i_type != X264GPU_MB_B_SKIP && e->cfg.dct8x8) {
        const int i_rd8 = rd_cost_b(a, i_type, i_partition, use16, 1, mb, lv);
        if (i_cost >= i_rd8) { if (i_cost > 0) i_satd_inter = (int)((int64_t)i_satd_inter * i_rd8 / i_cost); i_cost = i_rd8; t8 = 1; }
    }
    intra_rd(a, i_satd_inter * 17 / 16 + 1, mb, lv);
    int intra_type = -1;
    if (a->satd_i16 < i_cost) { i_cost = a->satd_i16; intra_type = X264GPU_MB_I16x16; }
    if (a->satd_i8 < i_cost) { i_cost = a->satd_i8; intra_type = X264GPU_MB_I8x8; }
    if (a->satd_i4 < i_cost) { i_cost = a->satd_i4; intra_type = X264GPU_MB_I4x4; }
    /* --subme 9 (i_mbrd 2 in B slices): the chosen intra type's modes once more on RD cost */
    if (intra_type >= 0 && a->mbrd >= 2 && ((e->cfg.rd >> 1) & 30)) intra_rd_refine(a, intra_type, mb, lv);
    rd_reset(a, mb, lv);
    if (intra_type >= 0) {
        mb->cost = i_cost;
        e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
        encode_intra_mb(a, intra_type, mb, lv);
        e->b_trellis = 0;
        e->intra_count++;
        return;
    }
    /* x264_refine_bidir (subme >= 5): the bi-predicted partitions of the winner refine both vectors together on SATD */
    if (a->subme >= 5 && i_type != X264GPU_MB_B_SKIP && i_type != X264GPU_MB_B_DIRECT) {
        if (i_partition == D_16x16) { if (use16 == 2) me_refine_bidir_satd(a, &a->bi16[0], &a->bi16[1], e->bipred_weight[a->bi16[0].ref][a->bi16[1].ref]); }
        else if (i_partition == D_16x8) { for (int i = 0; i < 2; i++) if (a->part16x8[i] == 2) me_refine_bidir_satd(a, &a->me16x8l[0][i], &a->me16x8l[1][i], e->bipred_weight[a->me16x8l[0][i].ref][a->me16x8l[1][i].ref]); }
        else if (i_partition == D_8x16) { for (int i = 0; i < 2; i++) if (a->part8x16[i] == 2) me_refine_bidir_satd(a, &a->me8x16l[0][i], &a->me8x16l[1][i], e->bipred_weight[a->me8x16l[0][i].ref][a->me8x16l[1][i].ref]); }
        else for (int i = 0; i < 4; i++) if (a->sub8[i] == 2) me_refine_bidir_satd(a, &a->me8l[0][i], &a->me8l[1][i], e->bipred_weight[a->me8l[0][i].ref][a->me8l[1][i].ref]);
    }
    /* --subme 9 (i_mbrd 2 in B slices): the chosen inter type's vectors once more on RD cost */
    if (a->mbrd >= 2 && ((e->cfg.rd >> 1) & 1) && (i_type == X264GPU_MB_B_INTER || i_type == X264GPU_MB_B_8x8)) {
        refine_inter_b_rd(a, i_type, i_partition, use16, t8, i_cost, mb, lv);
        rd_reset(a, mb, lv);
    }
    {
        const int bak = a->sub8[0];
        if (i_partition == D_16x16 && i_type == X264GPU_MB_B_INTER) a->sub8[0] = use16;
        fill_b_record(a, i_type, i_partition, mb);
        a->sub8[0] = bak;
    }
    a->force_t8 = t8;
    e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
    encode_inter_mb(a, mb, lv);
    e->b_trellis = 0;
    a->force_t8 = -1;
}


/* ===========================================================================================================================
 * RD refinement (x264 --subme 8 and up, i_mbrd >= 2: [x264-upstream] encoder/me.c x264_me_refine_qpel_rd, encoder/rdo.c x264_rd_cost_part /
 * rd_cost_i4x4 / rd_cost_i8x8 / rd_cost_chroma, encoder/analyse.c intra_rd_refine and the hooks in x264_macroblock_analyse;
 * oracle/RDREFINE_NOTES.md).  CABAC sessions.  Restated from memory like the rest of this file: parity unpinned.
 * Part costs carry 8 more bits than x264_rd_cost_mb's: (ssd << 8) + ((bits in 1/256 x lambda2 + 128) >> 8); they are only compared with each other. */
long x264o_cabac_part(x264o_cabac_ctx *c, int mbx, int mby, int kind, int a, int b, int d, const uint8_t *nnzc);

static int64_t part_bits(actx *a, int kind, int p0, int p1, int p2, int lambda2)
{
    uint8_t st[460];
    x264o_cabac_ctx cc;
    memcpy(st, a->e->cabac_state, sizeof(st));
    cabac_ctx_of(a, &cc, st);
    return ((int64_t)x264o_cabac_part(&cc, a->mbx, a->mby, kind, p0, p1, p2, a->nnzc) * lambda2 + 128) >> 8;
}

/* ssd_plane of the luma rectangle (x, y, w, h) of the macroblock: SSD + the psy-rd term (hadamard_ac from 8x8 up, SATD - SAD / 2 against zero below) */
static int ssd_luma_part(const actx *a, int x, int y, int w, int h)
{
    const x264o_encoder *e = a->e;
    const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + y) * e->fs + a->mbx * 16 + x;
    const pixel *rec = luma_plane((x264o_encoder *)e, e->cur, 0) + (size_t)(a->mby * 16 + y) * e->rs + a->mbx * 16 + x;
    int ssd = x264o_ssd(fenc, e->fs, rec, e->rs, w, h);
    if (e->cfg.psy_rd_q8) {
        int satd;
        if (w >= 8) {
            const uint64_t fdec_acs = x264o_hadamard_ac(rec, e->rs, w, h), fenc_acs = x264o_hadamard_ac(fenc, e->fs, w, h);
            satd = (abs((int32_t)fdec_acs - (int32_t)fenc_acs) + abs((int32_t)(fdec_acs >> 32) - (int32_t)(fenc_acs >> 32))) >> 1;
        } else {
            static const pixel zero[16] = { 0 };
            const int dc = x264o_sad(rec, e->rs, zero, 0, w, h) >> 1, fe = x264o_satd(fenc, e->fs, zero, 0, w, h) - (x264o_sad(fenc, e->fs, zero, 0, w, h) >> 1);
            satd = abs(x264o_satd(rec, e->rs, zero, 0, w, h) - dc - fe);
        }
        ssd += (satd * e->cfg.psy_rd_q8 * a->lambda + 128) >> 8;
    }
    return ssd;
}
static int ssd_chroma_part(const actx *a, int x, int y, int w, int h)      /* both planes of the NV12 rectangle, chroma samples */
{
    const x264o_encoder *e = a->e;
    const pixel *fuv = e->fenc_uv + (size_t)(a->mby * 8 + y) * e->fs + a->mbx * 16 + 2 * x;
    const pixel *ruv = chroma_plane((x264o_encoder *)e, e->cur) + (size_t)(a->mby * 8 + y) * e->rs + a->mbx * 16 + 2 * x;
    int s = 0;
    for (int j = 0; j < h; j++) for (int i = 0; i < 2 * w; i++) { const int d = fuv[j * e->fs + i] - ruv[j * e->rs + i]; s += d * d; }
    return s;
}

/* x264_macroblock_encode_p8x8 with b_skip_mc: the prediction of 8x8 block i8 lies in the reconstruction; luma with the macroblock's transform
 * size, the one chroma 4x4 block under it per plane without its DC */
static void encode_p8x8(actx *a, int i8, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int x8 = i8 & 1, y8 = i8 >> 1, qp = a->qp, qpc = a->qpc, b_decimate = e->cfg.dct_decimate || e->slice_type == X264GPU_SLICE_B;
    const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + 8 * y8) * e->fs + a->mbx * 16 + 8 * x8;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)(a->mby * 16 + 8 * y8) * e->rs + a->mbx * 16 + 8 * x8;
    mb->cbp_luma &= (uint8_t)~(1 << i8);
    mb->nnz &= ~(0xfu << (4 * i8));
    for (int b = 4 * i8; b < 4 * i8 + 4; b++) { memset(lv + b * 16, 0, 32); a->nnzc[b] = 0; }
    if (mb->transform8x8) {
        dctcoef d[64];
        int16_t scan[64];
        x264o_sub8x8_dct8(d, fenc, e->fs, rec, e->rs);
        int nz = quant_8x8(e, d, e->qt.quant8_mf[X264O_CQM_8PY][qp], e->qt.quant8_bias[X264O_CQM_8PY][qp], qp, 0);
        if (nz) {
            for (int k = 0; k < 64; k++) scan[k] = d[x264o_zigzag8[k]];
            if (b_decimate && !(e->b_trellis & TR_P8)) nz = x264o_decimate_score(scan, 64) >= 4;      /* (x264: b_decimate && !h->mb.b_trellis) */
        }
        if (nz) {
            for (int k = 0; k < 64; k++) { lv[(i8 * 4 + (k & 3)) * 16 + (k >> 2)] = scan[k]; if (scan[k]) mb->nnz |= 1u << (i8 * 4 + (k & 3)); }
            x264o_dequant_8x8(d, e->qt.dequant8_mf, qp);
            x264o_add8x8_idct8(rec, e->rs, d);
            mb->cbp_luma |= (uint8_t)(1 << i8);
            for (int b = 4 * i8; b < 4 * i8 + 4; b++) a->nnzc[b] = 1;
        }
    } else {
        dctcoef d[4][16];
        int nz[4], any = 0, score = b_decimate ? 0 : 4;
        for (int k = 0; k < 4; k++) {
            const int b = 4 * i8 + k, ox = (k & 1) * 4, oy = (k >> 1) * 4;
            x264o_sub4x4_dct(d[k], fenc + oy * e->fs + ox, e->fs, rec + oy * e->rs + ox, e->rs);
            nz[k] = quant_4x4(e, d[k], e->qt.quant4_mf[X264O_CQM_4PY][qp], e->qt.quant4_bias[X264O_CQM_4PY][qp], qp, 2, 0);
            a->nnzc[b] = (uint8_t)(nz[k] != 0);
            if (nz[k]) {
                scan4(lv + b * 16, d[k]);
                x264o_dequant_4x4(d[k], e->qt.dequant4_mf, qp);
                if (score < 4) score += x264o_decimate_score(lv + b * 16, 16);
                any = 1;
            }
        }
        if (any) {
            if (score < 4) for (int b = 4 * i8; b < 4 * i8 + 4; b++) { a->nnzc[b] = 0; memset(lv + b * 16, 0, 32); }
            else {
                for (int k = 0; k < 4; k++) if (nz[k]) { x264o_add4x4_idct(rec + (k >> 1) * 4 * e->rs + (k & 1) * 4, e->rs, d[k]); mb->nnz |= 1u << (4 * i8 + k); }
                mb->cbp_luma |= (uint8_t)(1 << i8);
            }
        }
    }
    const pixel *fuv = e->fenc_uv + (size_t)(a->mby * 8 + 4 * y8) * e->fs + a->mbx * 16 + 8 * x8;
    pixel *ruv = chroma_plane(e, e->cur) + (size_t)(a->mby * 8 + 4 * y8) * e->rs + a->mbx * 16 + 8 * x8;
    for (int c = 0; c < 2; c++) {
        pixel f[16], p[16];
        dctcoef d[16];
        int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i8) * 16;
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { f[y * 4 + x] = fuv[y * e->fs + 2 * x + c]; p[y * 4 + x] = ruv[y * e->rs + 2 * x + c]; }
        x264o_sub4x4_dct(d, f, 4, p, 4);
        d[0] = 0;
        const int nz = quant_4x4(e, d, e->qt.quant4_mf[X264O_CQM_4PC][qpc], e->qt.quant4_bias[X264O_CQM_4PC][qpc], qpc, 4, 0);
        a->nnzc[16 + c * 4 + i8] = (uint8_t)(nz != 0);
        mb->nnz &= ~(1u << (16 + c * 4 + i8));
        memset(l, 0, 32);
        if (nz) {
            scan4(l, d);
            x264o_dequant_4x4(d, e->qt.dequant4_mf, qpc);
            x264o_add4x4_idct(p, 4, d);
            for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) ruv[y * e->rs + 2 * x + c] = p[y * 4 + x];
            mb->nnz |= 1u << (16 + c * 4 + i8);
        }
    }
    mb->cbp_chroma = 2;
}

/* x264_rd_cost_part of an inter part (pixel: 1 16x8, 2 8x16, 3 8x8) whose prediction lies in the reconstruction */
static int64_t rd_cost_part(actx *a, int i8, int psize, int done, x264gpu_mb *mb, int16_t *lv)
{
    mb->cbp_luma = 0;
    encode_p8x8(a, i8, mb, lv);
    if (psize == 1) encode_p8x8(a, i8 + 1, mb, lv);
    if (psize == 2) encode_p8x8(a, i8 + 2, mb, lv);
    const int x = 8 * (i8 & 1), y = 8 * (i8 >> 1), w = psize == 1 ? 16 : 8, h = psize == 2 ? 16 : 8;
    int64_t ssd = ssd_luma_part(a, x, y, w, h);
    ssd += ((int64_t)ssd_chroma_part(a, x >> 1, y >> 1, w >> 1, h >> 1) * a->chroma_lambda2_offset + 128) >> 8;
    return (ssd << 8) + part_bits(a, a->e->slice_type == X264GPU_SLICE_B ? 4 : 0, i8, psize, done, a->lambda2);
}

/* x264_me_refine_qpel_rd of one part of the chosen P type: a sub-pel hexagon + square walk on RD cost, candidates gated by SATD.  mb holds
 * the macroblock as decided so far (type, partition, transform size, the vectors of the parts refined before); *done = their 8x8 blocks */
static void me_refine_qpel_rd(actx *a, me_t *m, int i8, int t8, int *done, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int psize = m->w == 16 ? (m->h == 16 ? 0 : 1) : (m->h == 16 ? 2 : 3);
    const int bx8 = i8 & 1, by8 = i8 >> 1, w8 = m->w >> 3, h8 = m->h >> 3;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)(a->mby * 16 + m->oy) * e->rs + a->mbx * 16 + m->ox;
    pixel *ruv = chroma_plane(e, e->cur) + (size_t)(a->mby * 8 + m->oy / 2) * e->rs + a->mbx * 16 + m->ox;
    const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + m->oy) * e->fs + a->mbx * 16 + m->ox;
    int64_t bcost = INT64_MAX;
    const int mv0x = m->mv[0], mv0y = m->mv[1];          /* (for 16x16 the candidates pass through a->me16, which IS m) */
    int bmx = mv0x, bmy = mv0y, omx, omy, pmx, pmy, satd, bsatd = COST_MAX, dir = -2;
    int last_mvd[2] = { 0, 0 }, priced = 0;
    const int bsl = e->slice_type == X264GPU_SLICE_B, l = bsl ? m->list : 0;      /* B slices: the part's list (x264_me_refine_qpel_rd's i_list) */
    int16_t (*rmv)[2] = l ? mb->mv1 : mb->mv;
    uint8_t *amv = l ? e->amvd1 : e->amvd;
    if (psize != 0 && i8 != 0) predict_mv_l(a, l, bx8, by8, w8, m->ref, m->mvp);      /* the parts refined before moved */
    pmx = m->mvp[0]; pmy = m->mvp[1];
    sctx S;
    sctx_init(&S, a, m);
#define SATD_THRESH(c) ((c) + ((c) >> 4))
#define COST_MV_SATD(mx, my, dst, avoid_mvp) do { \
        if (!(avoid_mvp) || !((mx) == pmx && (my) == pmy)) { \
            pixel pred_[256]; \
            get_ref(&S, pred_, mx, my); \
            for (int y_ = 0; y_ < m->h; y_++) memcpy(rec + y_ * e->rs, pred_ + y_ * 16, m->w); \
            dst = mbcmp(a, fenc, e->fs, pred_, 16, m->w, m->h) + S.cmx[mx] + S.cmy[my]; \
            if (dst < bsatd) bsatd = dst; \
        } else dst = COST_MAX; } while (0)
#define COST_MV_RD(mx, my, sat, do_dir, mdir) do { \
        if ((sat) <= SATD_THRESH(bsatd)) { \
            int64_t cost_; \
            for (int k_ = 0; k_ < 4; k_++) if ((k_ & 1) >= bx8 && (k_ & 1) < bx8 + w8 && (k_ >> 1) >= by8 && (k_ >> 1) < by8 + h8) { rmv[k_][0] = (int16_t)(mx); rmv[k_][1] = (int16_t)(my); } \
            if (psize == 0 && bsl) { \
                m->mv[0] = (mx); m->mv[1] = (my);          /* (m is a->me16l[l]: the candidate's record is filled from it) */ \
                cost_ = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x16, l, t8, mb, lv); \
            } else if (psize == 0) { \
                a->me16.mv[0] = (mx); a->me16.mv[1] = (my); \
                cost_ = rd_cost_inter(a, D_16x16, t8, mb, lv); \
            } else { \
                pixel pu_[64], pv_[64]; \
                x264o_mc_chroma(pu_, pv_, 8, chroma_plane(e, S.refslot), e->rs, a->mbx * 8 + m->ox / 2, a->mby * 8 + m->oy / 2, mx, my, m->w / 2, m->h / 2); \
                if (!bsl && e->wc0[m->ref].on[0]) x264o_mc_weight(pu_, 8, pu_, 8, m->w / 2, m->h / 2, e->wc0[m->ref].scale[0], e->wc0[m->ref].denom, e->wc0[m->ref].offset[0]); \
                if (!bsl && e->wc0[m->ref].on[1]) x264o_mc_weight(pv_, 8, pv_, 8, m->w / 2, m->h / 2, e->wc0[m->ref].scale[1], e->wc0[m->ref].denom, e->wc0[m->ref].offset[1]); \
                for (int y_ = 0; y_ < m->h / 2; y_++) for (int x_ = 0; x_ < m->w / 2; x_++) { ruv[y_ * e->rs + 2 * x_] = pu_[y_ * 8 + x_]; ruv[y_ * e->rs + 2 * x_ + 1] = pv_[y_ * 8 + x_]; } \
                cost_ = rd_cost_part(a, i8, psize, *done, mb, lv); \
            } \
            last_mvd[0] = abs((mx) - m->mvp[0]); last_mvd[1] = abs((my) - m->mvp[1]); priced = 1; \
            if (cost_ < bcost) { bcost = cost_; bmx = (mx); bmy = (my); dir = (do_dir) ? (mdir) : dir; } \
        } } while (0)
    COST_MV_SATD(bmx, bmy, bsatd, 0);
    if (psize != 0) COST_MV_RD(bmx, bmy, 0, 0, 0);
    else bcost = m->cost;
    /* the predicted vector */
    if ((bmx != pmx || bmy != pmy) && pmx >= a->smin[0] && pmx <= a->smax[0] && pmy >= a->smin[1] && pmy <= a->smax[1]) {
        COST_MV_SATD(pmx, pmy, satd, 0);
        COST_MV_RD(pmx, pmy, satd, 0, 0);
        /* the hexagon never repeats its centre: if the predictor won, the vector to avoid becomes the old one */
        if (bmx == pmx && bmy == pmy) { pmx = mv0x; pmy = mv0y; }
    }
    if (bmy < a->smin[1] + 3 || bmy > a->smax[1] - 3 || bmx < a->smin[0] + 3 || bmx > a->smax[0] - 3) {
        /* too close to the limits for the walk: the part keeps its vector; the |mvd| x264's size coder cached last stays (when it ran) */
        m->mv[0] = mv0x; m->mv[1] = mv0y;
        goto finish;
    }
    /* sub-pel hexagon, the pattern of the full-pel one */
    dir = -2; omx = bmx; omy = bmy;
    for (int j = 0; j < 6; j++) {
        COST_MV_SATD(omx + hex2[j + 1][0], omy + hex2[j + 1][1], satd, 1);
        COST_MV_RD(omx + hex2[j + 1][0], omy + hex2[j + 1][1], satd, 1, j);
    }
    if (dir != -2) {
        /* half hexagons that do not overlap the previous one */
        for (int i = 1; i < 10; i++) {
            const int odir = mod6m1[dir + 1];
            if (bmy < a->smin[1] + 3 || bmy > a->smax[1] - 3) break;
            dir = -2; omx = bmx; omy = bmy;
            for (int j = 0; j < 3; j++) {
                COST_MV_SATD(omx + hex2[odir + j][0], omy + hex2[odir + j][1], satd, 1);
                COST_MV_RD(omx + hex2[odir + j][0], omy + hex2[odir + j][1], satd, 1, odir - 1 + j);
            }
            if (dir == -2) break;
        }
    }
    /* square refine */
    omx = bmx; omy = bmy;
    for (int i = 0; i < 8; i++) {
        COST_MV_SATD(omx + square1[i + 1][0], omy + square1[i + 1][1], satd, 1);
        COST_MV_RD(omx + square1[i + 1][0], omy + square1[i + 1][1], satd, 1, 0);
    }
    m->cost = bcost > COST_MAX ? COST_MAX : (int)bcost;
    m->mv[0] = bmx; m->mv[1] = bmy;
    last_mvd[0] = abs(bmx - m->mvp[0]); last_mvd[1] = abs(bmy - m->mvp[1]); priced = 1;
finish:
#undef COST_MV_SATD
#undef COST_MV_RD
#undef SATD_THRESH
    /* x264_macroblock_cache_mv / _mvd of the part: what the parts after it predict from and count their mvd contexts on */
    for (int k = 0; k < 4; k++)
        if ((k & 1) >= bx8 && (k & 1) < bx8 + w8 && (k >> 1) >= by8 && (k >> 1) < by8 + h8) {
            rmv[k][0] = (int16_t)m->mv[0]; rmv[k][1] = (int16_t)m->mv[1];
            if (priced) { amv[((size_t)a->mi * 4 + k) * 2] = (uint8_t)(last_mvd[0] < 66 ? last_mvd[0] : 66); amv[((size_t)a->mi * 4 + k) * 2 + 1] = (uint8_t)(last_mvd[1] < 66 ? last_mvd[1] : 66); }
            *done |= 1 << k;
        }
    cache_block_l(a, l, bx8, by8, w8, h8, m->ref, m->mv);
}

/* x264_me_refine_bidir_rd ([x264-upstream] encoder/me.c me_refine_bidir with rd = 1): both vectors of a bi-predicted part walk together, up to two
 * components at a time, up to eight rounds; a pair whose SATD cost is within 17/16 of the best SATD so far is priced on RD cost (the whole macroblock
 * for 16x16, else the part), and the best RD cost moves the centre.  The part's final vectors and |mvd| (capped at 33 here) go into the caches. */
static void me_refine_bidir_rd(actx *a, me_t *m0, me_t *m1, int i_weight, int i8, int t8, int *done, x264gpu_mb *mb, int16_t *lv)
{
    static const int8_t dia4d[33][4] = {
        { 0, 0, 0, 0 },
        { 0, 0, 0, 1 }, { 0, 0, 0, -1 }, { 0, 0, 1, 0 }, { 0, 0, -1, 0 }, { 0, 1, 0, 0 }, { 0, -1, 0, 0 }, { 1, 0, 0, 0 }, { -1, 0, 0, 0 },
        { 0, 0, 1, 1 }, { 0, 0, -1, -1 }, { 0, 1, 1, 0 }, { 0, -1, -1, 0 }, { 1, 1, 0, 0 }, { -1, -1, 0, 0 }, { 1, 0, 0, 1 }, { -1, 0, 0, -1 },
        { 0, 1, 0, 1 }, { 0, -1, 0, -1 }, { 1, 0, 1, 0 }, { -1, 0, -1, 0 }, { 0, 0, -1, 1 }, { 0, 0, 1, -1 }, { 0, -1, 1, 0 }, { 0, 1, -1, 0 },
        { -1, 1, 0, 0 }, { 1, -1, 0, 0 }, { 1, 0, 0, -1 }, { -1, 0, 0, 1 }, { 0, -1, 0, 1 }, { 0, 1, 0, -1 }, { -1, 0, 1, 0 }, { 1, 0, -1, 0 } };
    x264o_encoder *e = a->e;
    const int bw = m0->w, bh = m0->h, psize = bw == 16 ? (bh == 16 ? 0 : 1) : (bh == 16 ? 2 : 3);
    const int bx8 = i8 & 1, by8 = i8 >> 1, w8 = bw >> 3, h8 = bh >> 3;
    int bm0x = m0->mv[0], bm0y = m0->mv[1], bm1x = m1->mv[0], bm1y = m1->mv[1], bcost = COST_MAX, mc_list0 = 1, mc_list1 = 1;
    int64_t bcostrd = INT64_MAX;
    if (bm0y < a->smin[1] + 8 || bm1y < a->smin[1] + 8 || bm0y > a->smax[1] - 8 || bm1y > a->smax[1] - 8 ||
        bm0x < a->smin[0] + 8 || bm1x < a->smin[0] + 8 || bm0x > a->smax[0] - 8 || bm1x > a->smax[0] - 8) {
        /* too close to the limits: the part keeps its vectors (and whatever |mvd| the caches held: zero here) */
        for (int k = 0; k < 4; k++) if ((k & 1) >= bx8 && (k & 1) < bx8 + w8 && (k >> 1) >= by8 && (k >> 1) < by8 + h8) *done |= 1 << k;
        cache_block_l(a, 0, bx8, by8, w8, h8, m0->ref, m0->mv); cache_block_l(a, 1, bx8, by8, w8, h8, m1->ref, m1->mv);
        return;
    }
    if (psize != 0 && i8 != 0) { predict_mv_l(a, 0, bx8, by8, w8, m0->ref, m0->mvp); predict_mv_l(a, 1, bx8, by8, w8, m1->ref, m1->mvp); }
    const uint16_t *cm0x = a->cost_mv - m0->mvp[0], *cm0y = a->cost_mv - m0->mvp[1], *cm1x = a->cost_mv - m1->mvp[0], *cm1y = a->cost_mv - m1->mvp[1];
    const pixel *fenc = e->fenc_y + (size_t)(a->mby * 16 + m0->oy) * e->fs + a->mbx * 16 + m0->ox;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)(a->mby * 16 + m0->oy) * e->rs + a->mbx * 16 + m0->ox;
    pixel *ruv = chroma_plane(e, e->cur) + (size_t)(a->mby * 8 + m0->oy / 2) * e->rs + a->mbx * 16 + m0->ox;
    static pixel buf[2][9][256], bufu[2][9][64], bufv[2][9][64];
    uint8_t visited[8][8][8];
    memset(visited, 0, sizeof(visited));
    sctx S0, S1;
    sctx_init(&S0, a, m0); sctx_init(&S1, a, m1);
    for (int pass = 0; pass < 8; pass++) {
        int bestj = 0;
        for (int li = 0; li < 2; li++) {
            if (!(li ? mc_list1 : mc_list0)) continue;
            const sctx *S = li ? &S1 : &S0;
            const me_t *m = li ? m1 : m0;
            for (int j = 0; j < 9; j++) {
                const int q = 4 + 3 * square1[j][0] + square1[j][1], mx = (li ? bm1x : bm0x) + square1[j][0], my = (li ? bm1y : bm0y) + square1[j][1];
                get_ref(S, buf[li][q], mx, my);
                x264o_mc_chroma(bufu[li][q], bufv[li][q], 8, chroma_plane(e, S->refslot), e->rs, a->mbx * 8 + m->ox / 2, a->mby * 8 + m->oy / 2, mx, my, bw / 2, bh / 2);
            }
        }
        for (int j = !!pass; j < 33; j++) {
            const int m0x = dia4d[j][0] + bm0x, m0y = dia4d[j][1] + bm0y, m1x = dia4d[j][2] + bm1x, m1y = dia4d[j][3] + bm1y;
            if (pass && (visited[m0x & 7][m0y & 7][m1x & 7] & (1 << (m1y & 7)))) continue;
            const int i0 = 4 + 3 * dia4d[j][0] + dia4d[j][1], i1 = 4 + 3 * dia4d[j][2] + dia4d[j][3];
            pixel pix[256];
            visited[m0x & 7][m0y & 7][m1x & 7] |= (uint8_t)(1 << (m1y & 7));
            x264o_pixel_avg_weight(pix, 16, buf[0][i0], 16, buf[1][i1], 16, bw, bh, i_weight);
            const int cost = mbcmp(a, fenc, e->fs, pix, 16, bw, bh) + cm0x[m0x] + cm0y[m0y] + cm1x[m1x] + cm1y[m1y];
            if (cost < bcost + (bcost >> 4)) {
                int64_t costrd;
                if (cost < bcost) bcost = cost;
                for (int k = 0; k < 4; k++) if ((k & 1) >= bx8 && (k & 1) < bx8 + w8 && (k >> 1) >= by8 && (k >> 1) < by8 + h8) {
                    mb->mv[k][0] = (int16_t)m0x; mb->mv[k][1] = (int16_t)m0y; mb->mv1[k][0] = (int16_t)m1x; mb->mv1[k][1] = (int16_t)m1y;
                }
                if (psize == 0) {
                    const int s0x = m0->mv[0], s0y = m0->mv[1], s1x = m1->mv[0], s1y = m1->mv[1];
                    m0->mv[0] = m0x; m0->mv[1] = m0y; m1->mv[0] = m1x; m1->mv[1] = m1y;          /* (m0 / m1 are a->bi16[]: the candidate's record is filled from them) */
                    costrd = rd_cost_b(a, X264GPU_MB_B_INTER, D_16x16, 2, t8, mb, lv);
                    m0->mv[0] = s0x; m0->mv[1] = s0y; m1->mv[0] = s1x; m1->mv[1] = s1y;
                } else {
                    pixel pu[64], pv[64];
                    for (int y = 0; y < bh; y++) memcpy(rec + y * e->rs, pix + y * 16, bw);
                    x264o_pixel_avg_weight(pu, 8, bufu[0][i0], 8, bufu[1][i1], 8, bw / 2, bh / 2, i_weight);
                    x264o_pixel_avg_weight(pv, 8, bufv[0][i0], 8, bufv[1][i1], 8, bw / 2, bh / 2, i_weight);
                    for (int y = 0; y < bh / 2; y++) for (int x = 0; x < bw / 2; x++) { ruv[y * e->rs + 2 * x] = pu[y * 8 + x]; ruv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
                    costrd = rd_cost_part(a, i8, psize, *done, mb, lv);
                }
                if (costrd < bcostrd) { bcostrd = costrd; bestj = j; }
            }
        }
        if (!bestj) break;
        bm0x += dia4d[bestj][0]; bm0y += dia4d[bestj][1]; bm1x += dia4d[bestj][2]; bm1y += dia4d[bestj][3];
        mc_list0 = dia4d[bestj][0] | dia4d[bestj][1]; mc_list1 = dia4d[bestj][2] | dia4d[bestj][3];
    }
    m0->mv[0] = bm0x; m0->mv[1] = bm0y; m1->mv[0] = bm1x; m1->mv[1] = bm1y;
    for (int k = 0; k < 4; k++)
        if ((k & 1) >= bx8 && (k & 1) < bx8 + w8 && (k >> 1) >= by8 && (k >> 1) < by8 + h8) {
            const int d0x = abs(bm0x - m0->mvp[0]), d0y = abs(bm0y - m0->mvp[1]), d1x = abs(bm1x - m1->mvp[0]), d1y = abs(bm1y - m1->mvp[1]);
            mb->mv[k][0] = (int16_t)bm0x; mb->mv[k][1] = (int16_t)bm0y; mb->mv1[k][0] = (int16_t)bm1x; mb->mv1[k][1] = (int16_t)bm1y;
            e->amvd[((size_t)a->mi * 4 + k) * 2] = (uint8_t)(d0x < 33 ? d0x : 33); e->amvd[((size_t)a->mi * 4 + k) * 2 + 1] = (uint8_t)(d0y < 33 ? d0y : 33);
            e->amvd1[((size_t)a->mi * 4 + k) * 2] = (uint8_t)(d1x < 33 ? d1x : 33); e->amvd1[((size_t)a->mi * 4 + k) * 2 + 1] = (uint8_t)(d1y < 33 ? d1y : 33);
            *done |= 1 << k;
        }
    cache_block_l(a, 0, bx8, by8, w8, h8, m0->ref, m0->mv); cache_block_l(a, 1, bx8, by8, w8, h8, m1->ref, m1->mv);
}

/* the hook of x264_macroblock_analyse for a B inter type (B_L0_L0 .. B_8x8) chosen on RD cost at i_mbrd >= 2 (--subme 9 in B slices): every part
 * that uses one list goes through x264_me_refine_qpel_rd in that list, every bi-predicted part through x264_me_refine_bidir_rd; direct 8x8 blocks stay */
static void refine_inter_b_rd(actx *a, int i_type, int i_partition, int use16, int t8, int i_cost, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    int done = 0;
    rd_reset(a, mb, lv);
    {
        const int bak = a->sub8[0];
        if (i_partition == D_16x16 && i_type == X264GPU_MB_B_INTER) a->sub8[0] = use16;
        fill_b_record(a, i_type, i_partition, mb);
        a->sub8[0] = bak;
    }
    mb->transform8x8 = (uint8_t)t8;
    a->partition = i_partition; a->cur_valid = a->cur_valid1 = 0;
    memset(e->amvd + (size_t)a->mi * 8, 0, 8); memset(e->amvd1 + (size_t)a->mi * 8, 0, 8);
    const int np = i_partition == D_16x16 ? 1 : i_partition == D_8x8 ? 4 : 2;
    for (int i = 0; i < np; i++) {
        const int i8 = i_partition == D_16x8 ? 2 * i : i;
        const int use = i_partition == D_16x16 ? use16 : i_partition == D_16x8 ? a->part16x8[i] : i_partition == D_8x16 ? a->part8x16[i] : a->sub8[i];
        me_t *m0 = i_partition == D_16x16 ? (use == 2 ? &a->bi16[0] : &a->me16l[0]) : i_partition == D_16x8 ? &a->me16x8l[0][i] : i_partition == D_8x16 ? &a->me8x16l[0][i] : &a->me8l[0][i];
        me_t *m1 = i_partition == D_16x16 ? (use == 2 ? &a->bi16[1] : &a->me16l[1]) : i_partition == D_16x8 ? &a->me16x8l[1][i] : i_partition == D_8x16 ? &a->me8x16l[1][i] : &a->me8l[1][i];
        const int bx8 = i8 & 1, by8 = i8 >> 1, w8 = m0->w >> 3, h8 = m0->h >> 3;
        if (use == 3) {          /* a direct 8x8 block: nothing to refine; the blocks after it predict from its direct motion */
            cache_b_block(a, bx8, by8, 1, 1, 3, NULL, NULL, i8);
            done |= 1 << i8;
        } else if (use == 2) me_refine_bidir_rd(a, m0, m1, e->bipred_weight[m0->ref][m1->ref], i8, t8, &done, mb, lv);
        else {
            me_t *m = use ? m1 : m0;
            if (i_partition == D_16x16) m->cost = i_cost;
            me_refine_qpel_rd(a, m, i8, t8, &done, mb, lv);
            cache_block_l(a, 1 - use, bx8, by8, w8, h8, -1, NULL);          /* the list the part does not use: no reference */
        }
    }
}

/* the hook of x264_macroblock_analyse for an inter P type chosen on RD cost; returns the partition (a pair of halves refined onto one vector is 16x16) */
static int refine_inter_p_rd(actx *a, int i_partition, int t8, int i_cost, x264gpu_mb *mb, int16_t *lv)
{
    int done = 0;
    rd_reset(a, mb, lv);
    mb->type = i_partition == D_8x8 ? X264GPU_MB_P_8x8 : X264GPU_MB_P_L0; mb->partition = (uint8_t)i_partition; mb->transform8x8 = (uint8_t)t8;
    for (int k = 0; k < 4; k++) {
        const me_t *m = i_partition == D_16x16 ? &a->me16 : i_partition == D_16x8 ? &a->me16x8[k >> 1] : i_partition == D_8x16 ? &a->me8x16[k & 1] : &a->me8[k];
        mb->ref[k] = (int8_t)m->ref; mb->mv[k][0] = (int16_t)m->mv[0]; mb->mv[k][1] = (int16_t)m->mv[1];
    }
    a->partition = i_partition; a->cur_valid = 0;
    memset(a->e->amvd + (size_t)a->mi * 8, 0, 8);
    if (i_partition == D_16x16) { a->me16.cost = i_cost; me_refine_qpel_rd(a, &a->me16, 0, t8, &done, mb, lv); }
    else if (i_partition == D_16x8) { me_refine_qpel_rd(a, &a->me16x8[0], 0, t8, &done, mb, lv); me_refine_qpel_rd(a, &a->me16x8[1], 2, t8, &done, mb, lv); }
    else if (i_partition == D_8x16) { me_refine_qpel_rd(a, &a->me8x16[0], 0, t8, &done, mb, lv); me_refine_qpel_rd(a, &a->me8x16[1], 1, t8, &done, mb, lv); }
    else for (int i = 0; i < 4; i++) me_refine_qpel_rd(a, &a->me8[i], i, t8, &done, mb, lv);
    /* "in rare cases we can end up qpel-RDing our way back to a larger partition size": the two halves on one vector and reference */
    if (i_partition == D_16x8 || i_partition == D_8x16) {
        const me_t *m0 = i_partition == D_16x8 ? &a->me16x8[0] : &a->me8x16[0], *m1 = i_partition == D_16x8 ? &a->me16x8[1] : &a->me8x16[1];
        if (m0->mv[0] == m1->mv[0] && m0->mv[1] == m1->mv[1] && m0->ref == m1->ref) { a->me16 = *m0; a->me16.w = a->me16.h = 16; a->me16.ox = a->me16.oy = 0; i_partition = D_16x16; }
    }
    return i_partition;
}

static int64_t rd_cost_chroma(actx *a, int mode, int b_dct, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    if (b_dct) {
        mb->nnz &= ~0x06ff0000u; mb->cbp_chroma = 0;
        encode_chroma(e, e->fenc_uv + (size_t)a->mby * 8 * e->fs + a->mbx * 16, chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16, a->qpc, 0, mb, lv);
        for (int i = 0; i < 8; i++) a->nnzc[16 + i] = (uint8_t)(mb->cbp_chroma == 2 ? (mb->nnz >> (16 + i)) & 1 : 0);
    }
    const int64_t ssd = ssd_chroma_part(a, 0, 0, 8, 8);
    mb->chroma_mode = (uint8_t)(mode > I_PRED_CHROMA_P ? I_PRED_CHROMA_DC : mode);
    return (ssd << 8) + part_bits(a, 3, 0, 0, 0, x264o_lambda2(a->qpc));
}

/* intra_rd_refine: the chosen intra type's modes once more, on RD cost.  mb / lv hold the last whole-macroblock encode of intra_rd */
static void intra_rd_refine(actx *a, int type, x264gpu_mb *mb, int16_t *lv)
{
    x264o_encoder *e = a->e;
    const int left = a->mbx > 0, top = a->mby > e->row0;
    const pixel *fenc = e->fenc_y + (size_t)a->mby * 16 * e->fs + a->mbx * 16;
    pixel *rec = luma_plane(e, e->cur, 0) + (size_t)a->mby * 16 * e->rs + a->mbx * 16;
    const int sites = e->cfg.rd >> 1;
    if (type == X264GPU_MB_I16x16 && (sites & 2)) {
        const int old = a->pred16;
        const int thresh = a->b_early_terminate ? a->satd_i16_dir[old] * 9 / 8 : COST_MAX;
        int modes[4], n = 0, best = a->satd_i16, bestm = old;
        if (left && top) { modes[n++] = I_PRED_16x16_V; modes[n++] = I_PRED_16x16_H; modes[n++] = I_PRED_16x16_DC; modes[n++] = I_PRED_16x16_P; }
        else if (left) { modes[n++] = I_PRED_16x16_DC_LEFT; modes[n++] = I_PRED_16x16_H; }
        else if (top) { modes[n++] = I_PRED_16x16_DC_TOP; modes[n++] = I_PRED_16x16_V; }
        else modes[n++] = I_PRED_16x16_DC_128;
        for (int i = 0; i < n; i++) {
            const int m = modes[i];
            if (m == old || a->satd_i16_dir[m] > thresh || a->satd_i16_dir[m] >= COST_MAX) continue;      /* (a mode the analysis never costed: x264 reads a stale value there) */
            a->pred16 = m;
            const int c = rd_cost_intra(a, X264GPU_MB_I16x16, mb, lv);
            if (c < best) { best = c; bestm = m; }
        }
        a->pred16 = bestm;
    }
    /* the chroma mode (every intra type) */
    if (sites & 4) {
        int cm[4], cn = 0;
        if (left && top) { cm[cn++] = I_PRED_CHROMA_DC; cm[cn++] = I_PRED_CHROMA_H; cm[cn++] = I_PRED_CHROMA_V; cm[cn++] = I_PRED_CHROMA_P; }
        else if (left) { cm[cn++] = I_PRED_CHROMA_DC_LEFT; cm[cn++] = I_PRED_CHROMA_H; }
        else if (top) { cm[cn++] = I_PRED_CHROMA_DC_TOP; cm[cn++] = I_PRED_CHROMA_V; }
        if (cn > 1) {
            const int thresh = a->b_early_terminate ? a->satd_chroma * 5 / 4 : COST_MAX;
            int sorted[4], i_max = 0;
            for (int i = 0; i < cn; i++) if (a->satd_chroma_dir[cm[i]] < thresh && cm[i] != a->predc) sorted[i_max++] = cm[i];
            if (i_max > 0) {
                /* the last thing coded was intra_rd's candidate: pixels and levels of the current chroma mode still lie there */
                pixel *ruv = chroma_plane(e, e->cur) + (size_t)a->mby * 8 * e->rs + a->mbx * 16;
                int64_t best = rd_cost_chroma(a, a->predc, 0, mb, lv);
                pixel nu[9 * 9], nvv[9 * 9], pu[64], pv[64];
                /* neighbours of the chroma block: the reconstruction around it (row -1, column -1) */
                memset(nu, 128, sizeof(nu)); memset(nvv, 128, sizeof(nvv));
                for (int y = -1; y < 8; y++)
                    for (int x = -1; x < 8; x++) {
                        if (y >= 0 && x >= 0) continue;
                        if ((y < 0 && !top) || (x < 0 && !left)) continue;
                        nu[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x]; nvv[(y + 1) * 9 + x + 1] = ruv[y * e->rs + 2 * x + 1];
                    }
                for (int i = 0; i < i_max; i++) {
                    const int m = sorted[i];
                    x264o_predict_8x8c(pu, 8, nu + 10, 9, m);
                    x264o_predict_8x8c(pv, 8, nvv + 10, 9, m);
                    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { ruv[y * e->rs + 2 * x] = pu[y * 8 + x]; ruv[y * e->rs + 2 * x + 1] = pv[y * 8 + x]; }
                    /* once a mode without a residual has been found, the remaining ones are not transformed ("any mode with a residual will be worse") */
                    const int64_t c = rd_cost_chroma(a, m, mb->cbp_chroma != 0, mb, lv);
                    if (c < best) { best = c; a->predc = m; }
                }
            }
        }
    }
    if (type == X264GPU_MB_I4x4 && (sites & 8)) {
        for (int k = 0; k < 16; k++) mb->i4_mode[k] = (uint8_t)a->pred4[k];
        for (int idx = 0; idx < 16; idx++) {
            const pixel *f = fenc + blk_y[idx] * 4 * e->fs + blk_x[idx] * 4;
            pixel *r = rec + blk_y[idx] * 4 * e->rs + blk_x[idx] * 4;
            const int avail = i4_avail(a, idx);
            int64_t best = INT64_MAX;
            pixel bestpix[16], p4[16];
            int16_t bestlv[16];
            int bestnz = 0, bestm = a->pred4[idx];
            for (const int8_t *pm = mode4_available(avail); *pm >= 0; pm++) {
                const int m = *pm;
                x264o_predict_4x4(p4, 4, r, e->rs, real_mode4(m, avail), avail);
                for (int y = 0; y < 4; y++) memcpy(r + y * e->rs, p4 + y * 4, 4);
                const int nz = encode_i4x4(e, f, r, a->qp, lv + idx * 16);
                a->nnzc[idx] = (uint8_t)nz;
                mb->i4_mode[idx] = (uint8_t)m;
                const int64_t c = ((int64_t)ssd_luma_part(a, blk_x[idx] * 4, blk_y[idx] * 4, 4, 4) << 8) + part_bits(a, 1, idx, 0, 0, a->lambda2);
                if (best > c) {
                    best = c; bestm = m; bestnz = nz;
                    for (int y = 0; y < 4; y++) memcpy(bestpix + y * 4, r + y * e->rs, 4);
                    memcpy(bestlv, lv + idx * 16, 32);
                }
            }
            for (int y = 0; y < 4; y++) memcpy(r + y * e->rs, bestpix + y * 4, 4);
            memcpy(lv + idx * 16, bestlv, 32);
            a->nnzc[idx] = (uint8_t)bestnz;
            a->pred4[idx] = bestm; mb->i4_mode[idx] = (uint8_t)bestm;
        }
    } else if (type == X264GPU_MB_I8x8 && (sites & 16)) {
        for (int k = 0; k < 4; k++) memset(mb->i4_mode + 4 * k, a->pred8[k], 4);
        mb->transform8x8 = 1;
        for (int idx = 0; idx < 4; idx++) {
            const pixel *f = fenc + (idx >> 1) * 8 * e->fs + (idx & 1) * 8;
            pixel *r = rec + (idx >> 1) * 8 * e->rs + (idx & 1) * 8;
            const int avail = i8_avail(a, idx);
            const int thresh = a->b_early_terminate ? a->satd_i8_dir[idx][a->pred8[idx]] * 11 / 8 : COST_MAX;
            int64_t best = INT64_MAX;
            pixel edge[33], p8[64], bestpix[64];
            int16_t bestlv[64];
            uint8_t bestnnz[4] = { 0, 0, 0, 0 };
            int cbp_new = 0, bestm = a->pred8[idx];
            uint32_t bestbits = 0;
            x264o_predict_8x8_filter(r, e->rs, edge, avail);
            for (const int8_t *pm = mode4_available(avail); *pm >= 0; pm++) {
                const int m = *pm;
                if (a->satd_i8_dir[idx][m] > thresh) continue;
                mb->cbp_luma = (uint8_t)(a->cbp_i8 & ~(1 << idx));
                mb->nnz &= ~(0xfu << (4 * idx));
                for (int b = 4 * idx; b < 4 * idx + 4; b++) memset(lv + b * 16, 0, 32);
                x264o_predict_8x8(p8, 8, edge, real_mode4(m, avail));
                for (int y = 0; y < 8; y++) memcpy(r + y * e->rs, p8 + y * 8, 8);
                const int nz = encode_i8x8(e, f, r, a->qp, idx, lv, &mb->nnz);
                if (nz) mb->cbp_luma |= (uint8_t)(1 << idx);
                for (int b = 4 * idx; b < 4 * idx + 4; b++) a->nnzc[b] = (uint8_t)nz;
                memset(mb->i4_mode + 4 * idx, m, 4);
                const int64_t c = ((int64_t)ssd_luma_part(a, (idx & 1) * 8, (idx >> 1) * 8, 8, 8) << 8) + part_bits(a, 2, idx, 0, 0, a->lambda2);
                if (best > c) {
                    best = c; bestm = m; cbp_new = mb->cbp_luma; bestbits = mb->nnz & (0xfu << (4 * idx));
                    for (int y = 0; y < 8; y++) memcpy(bestpix + y * 8, r + y * e->rs, 8);
                    for (int b = 0; b < 4; b++) { memcpy(bestlv + b * 16, lv + (4 * idx + b) * 16, 32); bestnnz[b] = a->nnzc[4 * idx + b]; }
                }
            }
            a->cbp_i8 = cbp_new;
            for (int y = 0; y < 8; y++) memcpy(r + y * e->rs, bestpix + y * 8, 8);
            for (int b = 0; b < 4; b++) { memcpy(lv + (4 * idx + b) * 16, bestlv + b * 16, 32); a->nnzc[4 * idx + b] = bestnnz[b]; }
            mb->nnz = (mb->nnz & ~(0xfu << (4 * idx))) | bestbits; mb->cbp_luma = (uint8_t)cbp_new;
            a->pred8[idx] = bestm; memset(mb->i4_mode + 4 * idx, bestm, 4);
        }
    }
}

static int mb_type_at(const x264o_encoder *e, int mbx, int mby)
{
    if (mbx < 0 || mby < e->row0 || mbx >= e->mbw) return -1;
    return e->mbs[mby * e->mbw + mbx].type;
}

static void macroblock_body(x264o_encoder *e, int mbx, int mby, actx *a)
{
    memset(a, 0, sizeof(*a));
    a->e = e; a->mbx = mbx; a->mby = mby; a->mi = mby * e->mbw + mbx;
    x264gpu_mb *mb = &e->mbs[a->mi];
    int16_t *lv = e->levels + (size_t)a->mi * X264GPU_MB_LEVELS;
    memset(mb, 0, sizeof(*mb));
    memset(lv, 0, X264GPU_MB_LEVELS * sizeof(int16_t));
    a->qp = e->mbqp[a->mi];
    /* x264_macroblock_analyse: under AQ a quantiser within 1 of the previous macroblock's becomes that one (a cheaper mb_qp_delta); QPRD
     * (subme >= 10, not built) would switch this off */
    if (e->cfg.aq_mode && abs(a->qp - e->last_qp) == 1) a->qp = e->last_qp;
    a->qpc = x264o_chroma_qp[clampi(a->qp + e->cfg.chroma_qp_offset, 0, 51)];
    a->lambda = x264o_lambda(a->qp);
    a->subme = clampi(e->cfg.subme, 0, 11); a->satd = a->subme > 1;
    a->nref = e->nref;
    a->type_left = mb_type_at(e, mbx - 1, mby); a->type_top = mb_type_at(e, mbx, mby - 1);
    a->type_tl = mb_type_at(e, mbx - 1, mby - 1); a->type_tr = mby > e->row0 ? mb_type_at(e, mbx + 1, mby - 1) : -1;
    a->satd_i16 = a->satd_i8 = a->satd_i4 = a->satd_chroma = COST_MAX;
    a->b_early_terminate = a->subme < 11;
    mb->qp = (uint8_t)a->qp;
    /* x264_macroblock_thread_init / mb_analyse_init: B slices analyse one sub-pel level down (6 -> 5, 8 -> 7); i_mbrd = (subme >= 6) + (subme >= 8) */
    if (e->slice_type == X264GPU_SLICE_B && (a->subme == 6 || a->subme == 8)) a->subme--;
    /* cfg.rd: bit 0 = RD mode decision; bits 1..5 = the sites of the RD refinement (x264's subme 8 = all five) — brought up on the device site by site */
    a->mbrd = e->cfg.rd ? 1 + (e->cfg.cabac && a->subme >= 8 && ((e->cfg.rd >> 1) & 31)) : 0;      /* (bit 6 of cfg.rd: deblock-aware RD, not a refinement site) */
    if (e->slice_type == X264GPU_SLICE_B && a->subme < 6) a->mbrd = 0;         /* (x264: i_mbrd from subme - 1 in B slices: no RD below --subme 7) */
    a->rd16x16 = COST_MAX; a->force_t8 = -1; a->cost8x8 = a->cost16x8 = a->cost8x16 = COST_MAX;
    for (int i = 0; i < 7; i++) a->satd_i16_dir[i] = a->satd_chroma_dir[i] = COST_MAX;
    for (int i = 0; i < 4; i++) for (int m = 0; m < 12; m++) a->satd_i8_dir[i][m] = COST_MAX;
    a->lambda2 = x264o_lambda2(a->qp);
    {   /* h->mb.i_chroma_lambda2_offset: 256 * 2^((qp - chroma qp) / 3) under psy, else 256 */
        static const uint16_t tab[37] = { 16, 20, 25, 32, 40, 50, 64, 80, 101, 128, 161, 203, 256, 322, 406, 512, 645, 812, 1024, 1290, 1625, 2048, 2580, 3250, 4096,
                                          5160, 6501, 8192, 10321, 13003, 16384, 20642, 26007, 32768, 41285, 52015, 65535 };
        const int idx = a->qp - a->qpc + 12;
        a->chroma_lambda2_offset = e->cfg.psy ? tab[idx < 0 ? 0 : idx > 36 ? 36 : idx] : 256;
    }

    if (e->slice_type == X264GPU_SLICE_I) {
        analyse_intra(a, COST_MAX);
        if (a->mbrd) intra_rd(a, COST_MAX, mb, lv);
        int i_cost = a->satd_i16, type = X264GPU_MB_I16x16;
        if (a->satd_i4 < i_cost) { i_cost = a->satd_i4; type = X264GPU_MB_I4x4; }
        if (a->satd_i8 < i_cost) { i_cost = a->satd_i8; type = X264GPU_MB_I8x8; }
        if (a->mbrd >= 2 && ((e->cfg.rd >> 1) & 30)) intra_rd_refine(a, type, mb, lv);
        if (a->mbrd) rd_reset(a, mb, lv);
        mb->cost = i_cost;
        e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;          /* --trellis 1: the final encode only */
        encode_intra_mb(a, type, mb, lv);
        e->b_trellis = 0;
        e->intra_count++;
        return;
    }

    /* ---- P / B slice ---- */
    a->cost_mv = x264o_cost_mv_for(e, a->qp);
    /* x264_macroblock_thread_init: b_chroma_me = --chroma-me && (P slice && subme >= 5 || B slice && subme >= 9) */
    a->chroma_me = e->cfg.chroma_me && a->subme >= (e->slice_type == X264GPU_SLICE_B ? 9 : 5);
    {   /* motion vector limits (x264_analyse_init: mv_min / mv_max, _spel clipped to --mvrange, _fpel inside the padded picture) */
        const int fr = 4 * (e->cfg.mv_range > 0 ? e->cfg.mv_range : 512);
        a->mv_min[0] = 4 * (-16 * mbx - 24); a->mv_max[0] = 4 * (16 * (e->mbw - mbx - 1) + 24);
        a->mv_min[1] = 4 * (-16 * mby - 24); a->mv_max[1] = 4 * (16 * (e->mbh - mby - 1) + 24);
        for (int k = 0; k < 2; k++) {
            a->smin[k] = clampi(a->mv_min[k], -fr, fr - 1); a->smax[k] = clampi(a->mv_max[k], -fr, fr - 1);
            a->fmin[k] = (a->smin[k] >> 2) + 6; a->fmax[k] = (a->smax[k] >> 2) - 6;
        }
    }
    /* fast intra decision: intra is unlikely unless a neighbour, the co-located macroblock of reference 0 or a third of the
     * macroblocks coded so far are intra */
    const int mi_in_slice = a->mi - e->row0 * e->mbw;            /* h->mb.i_mb_xy - h->sh.i_first_mb */
    if (a->b_early_terminate && mi_in_slice > 4) {
        const int colo = e->slice_type == X264GPU_SLICE_P ? e->mbtype[ref_slot(e, 0)][a->mi] : -1;      /* the co-located type counts in P slices only */
        /* (x264: "always run in fast-intra mode for subme < 3") */
        if (!(a->subme > 2 && (is_intra_type(a->type_left) || is_intra_type(a->type_top) || is_intra_type(a->type_tl) || is_intra_type(a->type_tr) ||
              is_intra_type(colo) || mi_in_slice < 3 * e->intra_count))) a->b_fast_intra = 1;
    }
    if (e->slice_type == X264GPU_SLICE_B) { macroblock_b(a, mb, lv); return; }
    a->cur_valid = 0;
    predict_mv_pskip(a, a->pskip_mv);
    int b_skip = 0;
    if (e->cfg.fast_pskip) {
        if (a->subme >= 3) a->b_try_skip = 1;
        else if (a->type_left == X264GPU_MB_P_SKIP || a->type_top == X264GPU_MB_P_SKIP || a->type_tl == X264GPU_MB_P_SKIP || a->type_tr == X264GPU_MB_P_SKIP)
            b_skip = probe_pskip(a);
    }
    int16_t (*mvr0)[2] = e->mv16[e->cur];
    if (b_skip) {
        /* set up the vectors for future predictors */
        mvr0[a->mi][0] = mvr0[a->mi][1] = 0;
        for (int r = 1; r < a->nref; r++) e->mvr[r][a->mi][0] = e->mvr[r][a->mi][1] = 0;
    } else if (analyse_inter_p16x16(a)) {
        b_skip = 1;
        for (int r = 1; r < a->nref; r++) e->mvr[r][a->mi][0] = e->mvr[r][a->mi][1] = 0;
    }
    if (b_skip) {
        mb->type = X264GPU_MB_P_SKIP; mb->partition = D_16x16;
        for (int k = 0; k < 4; k++) { mb->ref[k] = 0; mb->mv[k][0] = (int16_t)a->pskip_mv[0]; mb->mv[k][1] = (int16_t)a->pskip_mv[1]; }
        /* the skip vector is motion-compensated clipped to the padded picture (same samples as the unclipped vector) */
        x264gpu_mb t = *mb;
        for (int k = 0; k < 4; k++) { t.mv[k][0] = (int16_t)clampi(a->pskip_mv[0], a->mv_min[0], a->mv_max[0]); t.mv[k][1] = (int16_t)clampi(a->pskip_mv[1], a->mv_min[1], a->mv_max[1]); }
        encode_inter_mb(a, &t, lv);
        return;
    }

    const int psub16 = e->cfg.partitions & 1;
    if (psub16) { if (e->cfg.mixed_refs) analyse_inter_p8x8_mixed_ref(a); else analyse_inter_p8x8(a); }
    /* best inter mode */
    int i_type = X264GPU_MB_P_L0, i_partition = D_16x16, i_cost = a->me16.cost;
    if (psub16 && (!a->b_early_terminate || a->cost8x8 < a->me16.cost)) { i_type = X264GPU_MB_P_8x8; i_partition = D_8x8; i_cost = a->cost8x8; }
    {
        const int i_thresh16x8 = psub16 ? a->me8[1].cost_mv + a->me8[2].cost_mv : 0;
        if (psub16 && (!a->b_early_terminate || a->cost8x8 < a->me16.cost + i_thresh16x8)) {
            int avg = (a->me8[2].cost_mv + a->me8[2].ref_cost + a->me8[3].cost_mv + a->me8[3].ref_cost + 1) >> 1;
            a->cost_est16x8_1 = a->satd8x8[2] + a->satd8x8[3] + avg;
            analyse_inter_p16x8(a, i_cost);
            if (a->cost16x8 < i_cost) { i_type = X264GPU_MB_P_L0; i_partition = D_16x8; i_cost = a->cost16x8; }
            avg = (a->me8[1].cost_mv + a->me8[1].ref_cost + a->me8[3].cost_mv + a->me8[3].ref_cost + 1) >> 1;
            a->cost_est8x16_1 = a->satd8x8[1] + a->satd8x8[3] + avg;
            analyse_inter_p8x16(a, i_cost);
            if (a->cost8x16 < i_cost) { i_type = X264GPU_MB_P_L0; i_partition = D_8x16; i_cost = a->cost8x16; }
        }
    }
    /* refine the winner's quarter-pel vectors (no RD: always, unless full-pel only; with RD only the levels above 7 refine, by RD) */
    if (a->subme && !a->mbrd) {
        if (i_partition == D_16x16) { a->partition = D_16x16; me_refine_qpel(a, &a->me16); i_cost = a->me16.cost; }
        else if (i_partition == D_16x8) { me_refine_qpel(a, &a->me16x8[0]); me_refine_qpel(a, &a->me16x8[1]); i_cost = a->me16x8[0].cost + a->me16x8[1].cost; }
        else if (i_partition == D_8x16) { me_refine_qpel(a, &a->me8x16[0]); me_refine_qpel(a, &a->me8x16[1]); i_cost = a->me8x16[0].cost + a->me8x16[1].cost; }
        else { i_cost = 0; for (int i = 0; i < 4; i++) { me_refine_qpel(a, &a->me8[i]); i_cost += a->me8[i].cost; } }
    }
    if (a->chroma_me) {
        analyse_intra_chroma(a);
        analyse_intra(a, i_cost - a->satd_chroma);
        a->satd_i16 += a->satd_chroma; a->satd_i8 += a->satd_chroma; a->satd_i4 += a->satd_chroma;
    } else analyse_intra(a, i_cost);
    mb->aux[0] = i_cost; mb->aux[2] = a->me16.cost;
    { int mn = a->satd_i16 < a->satd_i8 ? a->satd_i16 : a->satd_i8; if (a->satd_i4 < mn) mn = a->satd_i4; mb->aux[1] = mn; }
    int t8 = -1;
    const int aux0 = mb->aux[0], aux1 = mb->aux[1], aux2 = mb->aux[2];        /* the RD passes below reset the record */
    if (a->mbrd) {
        int i_satd_inter = i_cost, i_satd_intra = a->satd_i16 < a->satd_i8 ? a->satd_i16 : a->satd_i8;
        if (a->satd_i4 < i_satd_intra) i_satd_intra = a->satd_i4;
        analyse_p_rd(a, i_satd_inter < i_satd_intra ? i_satd_inter : i_satd_intra, mb, lv);
        i_type = X264GPU_MB_P_L0; i_partition = D_16x16; i_cost = a->rd16x16;
        if (a->cost16x8 < i_cost) { i_cost = a->cost16x8; i_partition = D_16x8; }
        if (a->cost8x16 < i_cost) { i_cost = a->cost8x16; i_partition = D_8x16; }
        if (a->cost8x8 < i_cost) { i_cost = a->cost8x8; i_partition = D_8x8; i_type = X264GPU_MB_P_8x8; }
        t8 = 0;
        if (i_cost < COST_MAX && e->cfg.dct8x8) {       /* x264_mb_analyse_transform_rd: the other transform size for the winner */
            const int i_rd8 = rd_cost_inter(a, i_partition, 1, mb, lv);
            if (i_cost >= i_rd8) { if (i_cost > 0) i_satd_inter = (int)((int64_t)i_satd_inter * i_rd8 / i_cost); i_cost = i_rd8; t8 = 1; }
        }
        intra_rd(a, i_satd_inter * 5 / 4 + 1, mb, lv);
    }
    if (a->satd_i16 < i_cost) { i_cost = a->satd_i16; i_type = X264GPU_MB_I16x16; }
    if (a->satd_i8 < i_cost) { i_cost = a->satd_i8; i_type = X264GPU_MB_I8x8; }
    if (a->satd_i4 < i_cost) { i_cost = a->satd_i4; i_type = X264GPU_MB_I4x4; }
    if (a->mbrd >= 2) {
        if (is_intra_type(i_type)) { if ((e->cfg.rd >> 1) & 30) intra_rd_refine(a, i_type, mb, lv); }
        else if ((e->cfg.rd >> 1) & 1) i_partition = refine_inter_p_rd(a, i_partition, t8, i_cost, mb, lv);
    }
    if (a->mbrd) { rd_reset(a, mb, lv); mb->aux[0] = aux0; mb->aux[1] = aux1; mb->aux[2] = aux2; }
    mb->cost = i_cost;

    if (is_intra_type(i_type)) {
        e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
        encode_intra_mb(a, i_type, mb, lv);
        e->b_trellis = 0;
        e->intra_count++;
        return;
    }
    mb->type = (uint8_t)i_type; mb->partition = (uint8_t)i_partition;
    for (int k = 0; k < 4; k++) {
        const me_t *m = i_partition == D_16x16 ? &a->me16 : i_partition == D_16x8 ? &a->me16x8[k >> 1] : i_partition == D_8x16 ? &a->me8x16[k & 1] : &a->me8[k];
        mb->ref[k] = (int8_t)m->ref; mb->mv[k][0] = (int16_t)m->mv[0]; mb->mv[k][1] = (int16_t)m->mv[1];
    }
    a->force_t8 = t8;                 /* RD: the transform size chosen by transform_rd; else SA8D vs SATD */
    e->b_trellis = e->cfg.cabac ? e->cfg.trellis & 63 : 0;
    encode_inter_mb(a, mb, lv);
    e->b_trellis = 0;
}

void x264o_macroblock(x264o_encoder *e, int mbx, int mby)
{
    actx A;
    /* --trellis 2 (cfg.trellis bit 6): h->mb.b_trellis is on for the whole analysis of RD sessions too — the block encodes inside the intra
     * analysis and every RD candidate are quantised by the search (x264 mb_analyse_init: b_trellis = i_trellis > 1 && i_mbrd) */
    e->b_trellis = (e->cfg.trellis & 64) && e->cfg.rd && e->cfg.cabac && !(e->slice_type == X264GPU_SLICE_B && e->cfg.subme < 7) ? e->cfg.trellis & 63 : 0;      /* (B slices below --subme 7: i_mbrd 0) */
    macroblock_body(e, mbx, mby, &A);
    e->b_trellis = 0;
    const x264gpu_mb *mb = &e->mbs[mby * e->mbw + mbx];
    /* diagnostics for the tests: what mb_bits_cavlc says the macroblock layer of the final macroblock takes (0 for P_SKIP: it lives in a run) */
    if (e->mb_bits && !e->cfg.cabac) e->mb_bits[A.mi] = mb->type == X264GPU_MB_P_SKIP || mb->type == X264GPU_MB_B_SKIP ? 0 : mb_bits_cavlc(&A, mb, e->levels + (size_t)A.mi * X264GPU_MB_LEVELS);
    /* CABAC RD sessions: the finished macroblock moves the slice's context states on, as its entropy coding will */
    if (e->cfg.cabac && (e->cfg.rd || e->cfg.trellis)) {
        x264o_cabac_ctx cc;
        if (e->mb_bits && mb->type != X264GPU_MB_P_SKIP) {          /* diagnostics: the size estimate of the final macroblock, 1/256 bits */
            uint8_t st[460];
            memcpy(st, e->cabac_state, sizeof(st));
            cabac_ctx_of(&A, &cc, st);
            e->mb_bits[A.mi] = (int)x264o_cabac_mb(&cc, mbx, mby, 1);
        }
        cabac_ctx_of(&A, &cc, e->cabac_state);
        x264o_cabac_mb(&cc, mbx, mby, 0);
        e->last_dqp = cc.last_dqp;
    }
    /* h->mb.i_last_qp for the mb_qp_delta bits of the RD costs: macroblocks that send a delta set it */
    if (mb->type != X264GPU_MB_P_SKIP && (mb->cbp_luma || mb->cbp_chroma || mb->type == X264GPU_MB_I16x16)) {
        if (!(mb->type == X264GPU_MB_I16x16 && !mb->cbp_luma && !mb->cbp_chroma && !((mb->nnz >> 24) & 1) && mb->qp > e->last_qp)) e->last_qp = mb->qp;
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------
 * The lookahead's use of the same search (slicetype.c; x264 slicetype_mb_cost -> x264_me_search on an 8x8 block of the half-resolution
 * planes, lowres_context_init: lambda of qp 12, me = min(hex, --me) and sub-pel level 4 when --subme > 1, else dia and level 2, no chroma).
 * `lo` is an x264o_encoder shell whose DPB slots hold half-resolution plane sets and whose fenc is plane 0 of the picture being costed;
 * (bx, by) = the 8x8 block, lim = { fmin x, fmax x, fmin y, fmax y, smin x, smax x, smin y, smax y } as slicetype_mb_cost sets them. */
int x264o_lowres_me_search(x264o_encoder *lo, int slot, int bx, int by, const int mvp[2], int (*mvc)[2], int i_mvc, const int lim[8], int param_subme, int mv[2], int *cost_mv)
{
    actx A, *a = &A;
    memset(a, 0, sizeof(*a));
    a->e = lo; a->mbx = bx >> 1; a->mby = by >> 1;
    a->qp = 12; a->lambda = x264o_lambda(12);
    a->subme = param_subme > 1 ? 4 : 2; a->satd = param_subme > 1;
    a->cost_mv = x264o_cost_mv_for(lo, 12);
    a->fmin[0] = lim[0]; a->fmax[0] = lim[1]; a->fmin[1] = lim[2]; a->fmax[1] = lim[3];
    a->smin[0] = lim[4]; a->smax[0] = lim[5]; a->smin[1] = lim[6]; a->smax[1] = lim[7];
    a->nref = 1;
    lo->lslot[0][0] = slot; lo->nref_l[0] = 1; lo->slice_type = X264GPU_SLICE_P;
    me_t m;
    memset(&m, 0, sizeof(m));
    m.w = m.h = 8; m.ox = 8 * (bx & 1); m.oy = 8 * (by & 1);
    m.mvp[0] = mvp[0]; m.mvp[1] = mvp[1];
    me_search_ref(a, &m, mvc, i_mvc, NULL);
    mv[0] = m.mv[0]; mv[1] = m.mv[1];
    if (cost_mv) *cost_mv = m.cost_mv;
    return m.cost;
}

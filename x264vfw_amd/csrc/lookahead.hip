// lookahead.hip — lookahead frame cost on the device (include/x264gpu.h x264gpu_lookahead_*).
// Restates oracle/lookahead.c bit-exactly ([x264-upstream] encoder/slicetype.c x264_slicetype_frame_cost / _mb_cost for the
// pair p0 = previous picture, b = p1 = new picture; reached from the reference at codec.c:1693).
//
//   k_la_lowres : the four half-resolution phase planes of the mod-16 expanded luma, written with their replicated borders in
//                 one pass (HBM streaming: reads W*H, writes 4 * (W/2 + 64) * (H/2 + 64)).
//   k_la_cost   : one wavefront per 2x2 group of 8x8 half-resolution blocks.  The four inter searches run side by side in the
//                 candidate-parallel partition layout of k_analyse.hip.h (search_parts, global-memory mode: per-block predictor and
//                 limits), the intra costs in the (block, 4x4, row) layout (8x8c DC/H/V/P) and in (mode pair, block, row) for the
//                 filtered Intra_8x8 modes 3..8.  Frame sums are one atomicAdd per wave and counter.
#include "k_analyse.hip.h"
#include "intra8.hip.h"
#include <new>
#include <string.h>
#include <math.h>

namespace x264gpu {

constexpr int LPAD = 32;

struct LaK {
    const uint8_t *i420; size_t i420_bytes; int w, h;       // S tightly packed I420 pictures (luma only is read)
    int bw, bh, lw, lh, ls; size_t lplane, lpic;            // blocks, half-resolution plane geometry (lpic = 4 planes)
    uint8_t *cur; const uint8_t *prev;                      // S x 4 padded planes
    int16_t *mv_cur; const int16_t *mv_prev;                // S x bw*bh x 2 (quarter-pel)
    int8_t *in_cur; const int8_t *in_prev;                  // S x bw*bh: 1 = vector valid
    const uint16_t *cost_mv;                                // 2*MVCOST_HALF entries, lambda of qp 12
    int me_range, subme, lambda, have_prev;
    int32_t *out, *blocks;                                  // S x 4; optional S x bw*bh x 4 (intra cost, best cost, packed vector, inter)
};

__device__ __forceinline__ int la_avg4(int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; }

__global__ __launch_bounds__(256) void k_la_lowres(LaK k)
{
    const int x4 = ((int)(blockIdx.x * 256 + threadIdx.x)) * 4 - LPAD, y = (int)blockIdx.y - LPAD, s = blockIdx.z;
    if (x4 >= k.lw + LPAD) return;
    const uint8_t *src = k.i420 + (size_t)s * k.i420_bytes;
    const int cy = min(max(y, 0), k.lh - 1), Y = 2 * cy;
    const uint8_t *r0 = src + (size_t)min(Y, k.h - 1) * k.w, *r1 = src + (size_t)min(Y + 1, k.h - 1) * k.w, *r2 = src + (size_t)min(Y + 2, k.h - 1) * k.w;
    uint32_t o[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int X = 2 * min(max(x4 + i, 0), k.lw - 1);
        const int x0 = min(X, k.w - 1), x1 = min(X + 1, k.w - 1), x2 = min(X + 2, k.w - 1);
        const int a00 = r0[x0], a01 = r0[x1], a02 = r0[x2], a10 = r1[x0], a11 = r1[x1], a12 = r1[x2], a20 = r2[x0], a21 = r2[x1], a22 = r2[x2];
        o[0] |= (uint32_t)la_avg4(a00, a10, a01, a11) << (8 * i);
        o[1] |= (uint32_t)la_avg4(a01, a11, a02, a12) << (8 * i);
        o[2] |= (uint32_t)la_avg4(a10, a20, a11, a21) << (8 * i);
        o[3] |= (uint32_t)la_avg4(a11, a21, a12, a22) << (8 * i);
    }
    uint8_t *d = k.cur + (size_t)s * k.lpic + (size_t)(y + LPAD) * k.ls + x4 + LPAD;
#pragma unroll
    for (int p = 0; p < 4; p++) *(uint32_t *)(d + p * k.lplane) = o[p];
}

__global__ __launch_bounds__(256) void k_la_cost(LaK k)
{
    __shared__ uint32_t s_sub[4][SubGeo<2>::DWORDS];
    __shared__ __attribute__((aligned(8))) uint8_t s_tab[9 * 64];
    __shared__ uint8_t s_cnb[4][4][CNB_SIZE];
    __shared__ uint8_t s_u8[4][4][U8_SIZE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, s = blockIdx.y;      // scalar wave index
    for (int i = threadIdx.x; i < 9 * 64; i += 256) s_tab[i] = c_pred8_table[i];
    __syncthreads();
    const int gw = (k.bw + 1) >> 1, gh = (k.bh + 1) >> 1, g = blockIdx.x * 4 + wave;
    if (g >= gw * gh) return;                           // wave-uniform; no block-wide barrier below
    const int gx = g % gw, gy = g / gw;
    const int B = lane >> 4;                            // this lane's 8x8 block: one per DPP row = one partition of search_parts
    const int bx = 2 * gx + (B & 1), by = 2 * gy + (B >> 1);
    const bool bvalid = bx < k.bw && by < k.bh;         // blocks beyond an odd edge compute on the padding and are dropped
    const int nb = k.bw * k.bh, bi = min(by, k.bh - 1) * k.bw + min(bx, k.bw - 1);
    const uint8_t *cur0 = k.cur + (size_t)s * k.lpic + (size_t)LPAD * k.ls + LPAD;
    const uint8_t *prev0 = k.prev + (size_t)s * k.lpic + (size_t)LPAD * k.ls + LPAD;
    const uint8_t *blk = cur0 + (size_t)by * 8 * k.ls + bx * 8;       // this block's top-left source sample

    // ---------------- inter cost: x264_me_search on the previous picture's planes ----------------
    int pcost = 1 << 28, mvx = 0, mvy = 0;
    if (k.have_prev) {
        const int16_t *pmv = k.mv_prev + (size_t)s * nb * 2;
        const int8_t *pin = k.in_prev + (size_t)s * nb;
        int mvp0, mvp1;
        {   // median of the previous field: left, top, top-right (top-left when there is no top-right), oracle lookahead.c
            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = 0, c1 = 0;
            const bool ia = bx > 0, ib = by > 0, ic = by > 0 && bx + 1 < k.bw;
            if (ia && pin[bi - 1]) { a0 = pmv[2 * (bi - 1)]; a1 = pmv[2 * (bi - 1) + 1]; }
            if (ib && pin[bi - k.bw]) { b0 = pmv[2 * (bi - k.bw)]; b1 = pmv[2 * (bi - k.bw) + 1]; }
            if (ic) { if (pin[bi - k.bw + 1]) { c0 = pmv[2 * (bi - k.bw + 1)]; c1 = pmv[2 * (bi - k.bw + 1) + 1]; } }
            else if (by > 0 && bx > 0 && pin[bi - k.bw - 1]) { c0 = pmv[2 * (bi - k.bw - 1)]; c1 = pmv[2 * (bi - k.bw - 1) + 1]; }
            if (!ib && ia) { mvp0 = a0; mvp1 = a1; }
            else { mvp0 = median3(a0, b0, c0); mvp1 = median3(a1, b1, c1); }
        }
        // SATD at the zero vector in the (block, 4x4, row) layout: a zero predictor with SATD < 64 skips the search
        const int ci = (lane >> 2) & 3, j = lane & 3, zo = ((ci >> 1) * 4 + j) * k.ls + (ci & 1) * 4;
        const uint32_t e4 = *(const uint32_t *)(blk + zo);
        const uint8_t *pblk = prev0 + (size_t)by * 8 * k.ls + bx * 8;
        const int satd0 = row16_sum(satd4_half(e4, *(const uint32_t *)(pblk + zo), lane));
        const bool skip = !(mvp0 | mvp1) && satd0 < 64;

        PartCtx pc;
        pc.win = nullptr; pc.wx0 = pc.wy0 = 0; pc.cx = pc.cy = nullptr; pc.cbx = pc.cby = 0;
        pc.p00 = prev0; pc.pb = k.lplane; pc.rs = k.ls; pc.px = 16 * gx; pc.py = 16 * gy; pc.zx = pc.zy = 0; pc.cz = 0;
        pc.smin0 = 4 * (-8 * bx - 12); pc.smax0 = 4 * (8 * (k.bw - bx - 1) + 12);
        pc.smin1 = 4 * (-8 * by - 12); pc.smax1 = 4 * (8 * (k.bh - by - 1) + 12);
        pc.fmin0 = (pc.smin0 >> 2) + 6; pc.fmax0 = (pc.smax0 >> 2) - 6; pc.fmin1 = (pc.smin1 >> 2) + 6; pc.fmax1 = (pc.smax1 >> 2) - 6;
        pc.me_range = k.me_range; pc.me_method = 1; pc.hp_it = 1; pc.qp_it = 1; pc.lane = lane; pc.sub = s_sub[wave];
        pc.fenc = cur0 + (size_t)gy * 16 * k.ls + gx * 16; pc.fs = k.ls;
        pc.cref = nullptr; pc.fuv = nullptr; pc.chroma_me = 0; pc.csub = nullptr;
        pc.gcx = k.cost_mv + MVCOST_HALF - mvp0; pc.gcy = k.cost_mv + MVCOST_HALF - mvp1; pc.mvp0 = mvp0; pc.mvp1 = mvp1;
        pc.col_x = pmv[2 * bi]; pc.col_y = pmv[2 * bi + 1]; pc.has_col = pin[bi] != 0; pc.la_mode = true;
        int smx, smy;
        const int sc = search_parts<2, true>(pc, 3, 0, 0, smx, smy);
        if (skip) pcost = satd0;
        else {
            pcost = sc - (int)k.cost_mv[MVCOST_HALF];            // "remove mvcost from skip mbs"
            if (smx | smy) pcost += 5 * k.lambda;
            mvx = smx; mvy = smy;
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---------------- intra cost on source neighbours ----------------
    int icost;
    {
        const int ci = (lane >> 2) & 3, j = lane & 3, t = lane & 15;
        uint8_t *cnb = s_cnb[wave][B];
        if (t < 9) cnb[CNB_TOP - 1 + t] = blk[-(long)k.ls - 1 + t];             // corner + top[0..7]
        if (t < 8) cnb[CNB_LEFT + t] = blk[(long)t * k.ls - 1];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const PredC pp = predc_setup(cnb);
        const uint32_t e4 = *(const uint32_t *)(blk + ((ci >> 1) * 4 + j) * k.ls + (ci & 1) * 4);
        int best = 1 << 28;
        const int nm = k.subme > 1 ? 4 : 3;
        for (int m = 0; m < nm; m++)                                             // PREDC_DC, _H, _V, _P
            best = min(best, row16_sum(satd4_half(e4, predc_row4(cnb, pp, m, ci, j), lane)));
        if (k.subme > 1) {
            // Intra_8x8 modes 3..8 on the filtered edge (all neighbours exist: the planes are padded)
            for (int b = 0; b < 4; b++) {
                const uint8_t *bb = cur0 + (size_t)(2 * gy + (b >> 1)) * 8 * k.ls + (2 * gx + (b & 1)) * 8;
                pred8_build_u(s_u8[wave][b], bb, k.ls, AVAIL_LEFT | AVAIL_TOP | AVAIL_TOPRIGHT | AVAIL_TOPLEFT, lane);
            }
            const int g2 = lane >> 5, b8 = (lane >> 3) & 3, row = lane & 7;
            const uint8_t *rb = cur0 + (size_t)((2 * gy + (b8 >> 1)) * 8 + row) * k.ls + (2 * gx + (b8 & 1)) * 8;
            const uint2 er = *(const uint2 *)rb;
            int b88 = 1 << 28;
            for (int ps = 0; ps < 3; ps++) {
                uint32_t plo, phi;
                pred8_row8(s_u8[wave][b8], s_tab, 3 + 2 * ps + g2, row, plo, phi);
                int h = satd4_half(er.x, plo, lane) + satd4_half(er.y, phi, lane);
                h = quad_sum(h); h += xor4(h);
                b88 = min(b88, h);
            }
            b88 = min(b88, __shfl_xor(b88, 32));
            best = min(best, __shfl(b88, B * 8));
        }
        icost = best + 5 * k.lambda + 4;                                         // intra_penalty + lowres_penalty
    }

    // ---------------- decision, field for the next picture, frame sums ----------------
    const bool intra = !k.have_prev || icost < pcost;
    const int bcost = intra ? icost : pcost;
    const bool score = (bx > 0 && bx < k.bw - 1 && by > 0 && by < k.bh - 1) || k.bw <= 2 || k.bh <= 2;
    if ((lane & 15) == 0 && bvalid) {
        int16_t *mo = k.mv_cur + ((size_t)s * nb + bi) * 2;
        mo[0] = intra ? 0 : (int16_t)mvx; mo[1] = intra ? 0 : (int16_t)mvy;
        k.in_cur[(size_t)s * nb + bi] = intra ? 0 : 1;
        if (k.blocks) {
            int32_t *bo = k.blocks + ((size_t)s * nb + bi) * 4;
            bo[0] = icost; bo[1] = bcost; bo[2] = intra ? 0 : ((mvx & 0xffff) | (mvy << 16)); bo[3] = intra ? 0 : 1;
        }
    }
    const bool cnt = bvalid && score;
    int v0 = cnt ? icost : 0, v1 = cnt ? bcost : 0, v2 = cnt && intra && k.have_prev ? 1 : 0, v3 = cnt ? 1 : 0;
#define LA_SUM4(v) (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48))
    v0 = LA_SUM4(v0); v1 = LA_SUM4(v1); v2 = LA_SUM4(v2); v3 = LA_SUM4(v3);
#undef LA_SUM4
    if (lane == 0) {
        int32_t *o = k.out + (size_t)s * 4;
        atomicAdd(o, v0); atomicAdd(o + 1, v1); atomicAdd(o + 2, v2); atomicAdd(o + 3, v3);
    }
}

// ---- AQ offsets of a source picture (oracle x264o_aq_offsets): as k_aq, on the tight I420 input with clamped coordinates ----
__global__ __launch_bounds__(256) void k_la_aq(const uint8_t *__restrict__ i420, size_t i420_bytes, int w, int h, int bw, int nb, float strength, float *__restrict__ out, float *__restrict__ adj)
{
    const int lane = threadIdx.x & 63, r = lane & 15, s = blockIdx.y;
    const int bi = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool valid = bi < nb;
    const int b = valid ? bi : 0, bx = b % bw, by = b / bw;
    const uint8_t *Y = i420 + (size_t)s * i420_bytes, *U = Y + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    unsigned sum = 0, sqr = 0, su = 0, squ = 0, sv = 0, sqv = 0;
    const uint8_t *yr = Y + (size_t)min(by * 16 + r, h - 1) * w;
    for (int c = 0; c < 16; c++) { const unsigned p = yr[min(bx * 16 + c, w - 1)]; sum += p; sqr += p * p; }
    if (r < 8) {
        const size_t ro = (size_t)min(by * 8 + r, h / 2 - 1) * (w / 2);
        for (int c = 0; c < 8; c++) { const int x = min(bx * 8 + c, w / 2 - 1); const unsigned u = U[ro + x], v = V[ro + x]; su += u; squ += u * u; sv += v; sqv += v * v; }
    }
    sum = (unsigned)row16_sum((int)sum); sqr = (unsigned)row16_sum((int)sqr);
    su = (unsigned)row16_sum((int)su); squ = (unsigned)row16_sum((int)squ); sv = (unsigned)row16_sum((int)sv); sqv = (unsigned)row16_sum((int)sqv);
    const unsigned energy = (sqr - (sum * sum >> 8)) + (squ - (su * su >> 6)) + (sqv - (sv * sv >> 6));
    if (valid && r == 0) {
        if (adj) adj[(size_t)s * nb + bi] = sqrtf(sqrtf(sqrtf((float)energy + 1.f)));          // --aq-mode 2 / 3: (energy + 1)^(1/8), finished by k_la_aq_auto
        else out[(size_t)s * nb + bi] = f_mul(strength, f_sub(x264_log2(energy ? energy : 1u), 14.427f));          // x264_adaptive_quant_frame, mode 1
    }
}
// --aq-mode 2 / 3 (oracle x264o_aq_offsets_mode): the picture's mean and mean square of the per-macroblock values, summed in raster order by one
// thread a stream (x264's own order: float addition does not reassociate), then every macroblock's offset
__global__ __launch_bounds__(256) void k_la_aq_auto(const float *__restrict__ adj, int nb, int mode, float aqs, float *__restrict__ out)
{
    const int s = blockIdx.x;
    __shared__ float sh[2];
    const float *a = adj + (size_t)s * nb;
    if (threadIdx.x == 0) {
        float sum = 0.f, sq = 0.f;
        for (int i = 0; i < nb; i++) { const float q = a[i]; sum = f_add(sum, q); sq = f_add(sq, f_mul(q, q)); }
        sh[0] = f_div(sum, (float)nb); sh[1] = f_div(sq, (float)nb);
    }
    __syncthreads();
    const float mean = sh[0], strength = f_mul(aqs, mean);
    const float avg = f_sub(mean, f_div(f_mul(0.5f, f_sub(sh[1], 14.f)), mean));
    for (int i = threadIdx.x; i < nb; i += 256) {
        float q = f_mul(strength, f_sub(a[i], avg));
        if (mode == 3) q = f_add(q, f_mul(aqs, f_sub(1.f, f_div(14.f, f_mul(a[i], a[i])))));
        out[(size_t)s * nb + i] = q;
    }
}

// ---- macroblock-tree (oracle x264o_mbtree): one launch per picture walks its blocks and scatters the explained cost into the
// reference picture's accumulator (saturation is applied when an accumulator is read: the addends are non-negative) ----
__global__ __launch_bounds__(256) void k_mbtree_propagate(const int32_t *__restrict__ info, const float *__restrict__ aq, const int32_t *__restrict__ prop_in,
                                                           int32_t *__restrict__ prop_ref, int bw, int bh)
{
    const int nb = bw * bh, i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= nb) return;
    const int32_t *fi = info + ((size_t)s * nb + i) * 4;
    const int intra = min(fi[0], 16383), best = min(fi[1], 16383), inter = min(best, intra);
    if (!fi[3] || !intra) return;
    const int inv = x264_exp2fix8(aq ? aq[(size_t)s * nb + i] : 0.f);
    const int amount = x264_propagate_amount(min(prop_in[(size_t)s * nb + i], 32767), intra, inter, inv);          // mbtree_propagate_cost
    int32_t *ref = prop_ref + (size_t)s * nb;
    int x = (int)(short)(fi[2] & 0xffff), y = fi[2] >> 16;
    const int bx = i % bw, by = i / bw;
    if (!(x | y)) { atomicAdd(ref + i, amount); return; }
    const int mbx = (x >> 5) + bx, mby = (y >> 5) + by;
    x &= 31; y &= 31;
    const int w0 = ((32 - y) * (32 - x) * amount + 512) >> 10, w1 = ((32 - y) * x * amount + 512) >> 10;
    const int w2 = (y * (32 - x) * amount + 512) >> 10, w3 = (y * x * amount + 512) >> 10;
    if (mby >= 0 && mby < bh) { if (mbx >= 0 && mbx < bw) atomicAdd(ref + mby * bw + mbx, w0); if (mbx + 1 >= 0 && mbx + 1 < bw) atomicAdd(ref + mby * bw + mbx + 1, w1); }
    if (mby + 1 >= 0 && mby + 1 < bh) { if (mbx >= 0 && mbx < bw) atomicAdd(ref + (mby + 1) * bw + mbx, w2); if (mbx + 1 >= 0 && mbx + 1 < bw) atomicAdd(ref + (mby + 1) * bw + mbx + 1, w3); }
}

__global__ __launch_bounds__(256) void k_mbtree_finish(const int32_t *__restrict__ info, const float *__restrict__ aq, const int32_t *__restrict__ prop,
                                                        int nb, float strength, float *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= nb) return;
    const float a = aq ? aq[(size_t)s * nb + i] : 0.f;
    const int intra = (min(info[((size_t)s * nb + i) * 4], 16383) * x264_exp2fix8(a) + 128) >> 8;
    float off = a;
    if (intra) {
        const int p2 = min(prop[(size_t)s * nb + i], 32767) * 2;
        off = f_sub(a, f_mul(strength, f_add(f_sub(x264_log2((unsigned)(intra + p2)), x264_log2((unsigned)intra)), 0.f)));
    }
    out[(size_t)s * nb + i] = off;
}

}  // namespace x264gpu

using namespace x264gpu;

struct x264gpu_lookahead {
    int w, h, streams, bw, bh, lw, lh, ls, me_range, subme, lambda;
    size_t lplane, lpic;
    uint8_t *planes[2];
    int16_t *mv[2];
    int8_t *inter[2];
    uint16_t *cost_mv;
    int32_t *prop; int prop_cap;     // macroblock-tree accumulators: prop_cap pictures x streams x blocks
    float *aq_adj;                   // --aq-mode 2 / 3: the per-macroblock (energy + 1)^(1/8) of the picture in work
    int cur, have_prev;
};

extern "C" {

int x264gpu_lookahead_create(x264gpu_lookahead **out, int width, int height, int streams, int me_range, int subme)
{
    ARG_TRY(out && width >= 16 && height >= 16 && !(width & 1) && !(height & 1) && width <= 4096 && height <= 2304 && streams >= 1);
    x264gpu_lookahead *la = new (std::nothrow) x264gpu_lookahead();
    if (!la) return set_err(X264GPU_ENOMEM, "lookahead", hipSuccess);
    memset(la, 0, sizeof(*la));
    la->w = width; la->h = height; la->streams = streams;
    la->bw = (width + 15) / 16; la->bh = (height + 15) / 16;
    la->lw = la->bw * 8; la->lh = la->bh * 8;
    la->ls = (la->lw + 2 * LPAD + 63) / 64 * 64;
    la->lplane = (size_t)la->ls * (la->lh + 2 * LPAD); la->lpic = 4 * la->lplane;
    la->me_range = me_range < 4 ? 4 : me_range > 16 ? 16 : me_range; la->subme = subme;
    la->lambda = lambda_of(12);                                      // X264_LOOKAHEAD_QP
    const size_t S = streams, nb = (size_t)la->bw * la->bh;
    hipError_t er = hipSuccess;
    auto alloc = [&](void **p, size_t n) { if (er == hipSuccess) er = hipMalloc(p, n); if (er == hipSuccess) er = hipMemset(*p, 0, n); };
    for (int i = 0; i < 2; i++) {
        alloc((void **)&la->planes[i], S * la->lpic);
        alloc((void **)&la->mv[i], S * nb * 2 * sizeof(int16_t));
        alloc((void **)&la->inter[i], S * nb);
    }
    alloc((void **)&la->cost_mv, 2 * MVCOST_HALF * sizeof(uint16_t));
    if (er == hipSuccess) {
        uint16_t *hc = new (std::nothrow) uint16_t[2 * MVCOST_HALF];
        if (!hc) er = hipErrorOutOfMemory;
        else {
            for (int i = 0; i < MVCOST_HALF; i++) {
                float bits = log2f((float)(i + 1)) * 2.0f + 0.718f + (i ? 1.0f : 0.0f);
                int c = (int)((float)la->lambda * bits + 0.5f);
                if (c > 65535) c = 65535;
                hc[MVCOST_HALF + i] = (uint16_t)c; hc[MVCOST_HALF - i] = (uint16_t)c;
            }
            hc[0] = hc[1];
            er = hipMemcpy(la->cost_mv, hc, 2 * MVCOST_HALF * sizeof(uint16_t), hipMemcpyHostToDevice);
            delete[] hc;
        }
    }
    if (er != hipSuccess) { x264gpu_lookahead_destroy(la); return set_err(er == hipErrorOutOfMemory ? X264GPU_ENOMEM : X264GPU_EHIP, "lookahead buffers", er); }
    *out = la;
    return X264GPU_OK;
}

void x264gpu_lookahead_destroy(x264gpu_lookahead *la)
{
    if (!la) return;
    for (int i = 0; i < 2; i++) { (void)hipFree(la->planes[i]); (void)hipFree(la->mv[i]); (void)hipFree(la->inter[i]); }
    (void)hipFree(la->cost_mv); (void)hipFree(la->prop); (void)hipFree(la->aq_adj);
    delete la;
}

int x264gpu_lookahead_frame_cost(x264gpu_lookahead *la, const uint8_t *d_i420, int reset, int32_t *d_out, int32_t *d_blocks, void *stream)
{
    ARG_TRY(la && d_i420 && d_out);
    hipStream_t st = (hipStream_t)stream;
    if (reset) la->have_prev = 0;
    la->cur ^= 1;
    LaK k;
    memset(&k, 0, sizeof(k));
    k.i420 = d_i420; k.i420_bytes = (size_t)la->w * la->h * 3 / 2; k.w = la->w; k.h = la->h;
    k.bw = la->bw; k.bh = la->bh; k.lw = la->lw; k.lh = la->lh; k.ls = la->ls; k.lplane = la->lplane; k.lpic = la->lpic;
    k.cur = la->planes[la->cur]; k.prev = la->planes[la->cur ^ 1];
    k.mv_cur = la->mv[la->cur]; k.mv_prev = la->mv[la->cur ^ 1]; k.in_cur = la->inter[la->cur]; k.in_prev = la->inter[la->cur ^ 1];
    k.cost_mv = la->cost_mv; k.me_range = la->me_range; k.subme = la->subme; k.lambda = la->lambda; k.have_prev = la->have_prev;
    k.out = d_out; k.blocks = d_blocks;
    const int S = la->streams;
    HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)S * 4 * sizeof(int32_t), st));
    hipLaunchKernelGGL(k_la_lowres, dim3(((la->lw + 2 * LPAD) / 4 + 255) / 256, la->lh + 2 * LPAD, S), dim3(256), 0, st, k);
    const int groups = ((la->bw + 1) / 2) * ((la->bh + 1) / 2);
    hipLaunchKernelGGL(k_la_cost, dim3((groups + 3) / 4, S), dim3(256), 0, st, k);
    HIP_TRY(hipGetLastError());
    la->have_prev = 1;
    return X264GPU_OK;
}

int x264gpu_lookahead_aq_offsets(x264gpu_lookahead *la, const uint8_t *d_i420, float strength, float *d_out, void *stream)
{
    ARG_TRY(la && d_i420 && d_out);
    const int nb = la->bw * la->bh;
    hipLaunchKernelGGL(k_la_aq, dim3((nb + 15) / 16, la->streams), dim3(256), 0, (hipStream_t)stream, d_i420, (size_t)la->w * la->h * 3 / 2, la->w, la->h, la->bw, nb,
                       strength, d_out, (float *)nullptr);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}
int x264gpu_lookahead_aq_offsets_mode(x264gpu_lookahead *la, const uint8_t *d_i420, int mode, float strength, float *d_out, void *stream)
{
    ARG_TRY(la && d_i420 && d_out && mode >= 1 && mode <= 3);
    if (mode == 1) return x264gpu_lookahead_aq_offsets(la, d_i420, strength, d_out, stream);
    const int nb = la->bw * la->bh;
    if (!la->aq_adj) HIP_TRY(hipMalloc((void **)&la->aq_adj, (size_t)la->streams * nb * sizeof(float)));
    hipLaunchKernelGGL(k_la_aq, dim3((nb + 15) / 16, la->streams), dim3(256), 0, (hipStream_t)stream, d_i420, (size_t)la->w * la->h * 3 / 2, la->w, la->h, la->bw, nb,
                       strength, d_out, la->aq_adj);
    hipLaunchKernelGGL(k_la_aq_auto, dim3(la->streams), dim3(256), 0, (hipStream_t)stream, la->aq_adj, nb, mode, strength, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_lookahead_mbtree(x264gpu_lookahead *la, const int32_t *const *d_info, const float *const *d_aq, int n, float strength, float *d_out, void *stream)
{
    ARG_TRY(la && d_info && n >= 1 && n <= 256 && d_out);
    hipStream_t st = (hipStream_t)stream;
    const int nb = la->bw * la->bh, S = la->streams;
    const size_t per = (size_t)S * nb;
    if (la->prop_cap < n) {
        (void)hipFree(la->prop); la->prop = nullptr; la->prop_cap = 0;
        HIP_TRY(hipMalloc((void **)&la->prop, (size_t)n * per * sizeof(int32_t)));
        la->prop_cap = n;
    }
    HIP_TRY(hipMemsetAsync(la->prop, 0, (size_t)n * per * sizeof(int32_t), st));
    const dim3 grid((nb + 255) / 256, S);
    for (int j = n - 1; j >= 1; j--)
        hipLaunchKernelGGL(k_mbtree_propagate, grid, dim3(256), 0, st, d_info[j], d_aq ? d_aq[j] : nullptr, la->prop + (size_t)j * per, la->prop + (size_t)(j - 1) * per, la->bw, la->bh);
    hipLaunchKernelGGL(k_mbtree_finish, grid, dim3(256), 0, st, d_info[0], d_aq ? d_aq[0] : nullptr, la->prop, nb, strength, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

}  // extern "C"

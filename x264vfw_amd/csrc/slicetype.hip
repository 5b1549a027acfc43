// slicetype.hip — the lookahead's frame costs in x264's own structure on the device (include/x264gpu.h
// x264gpu_slicetype_*).  Restates oracle/slicetype.c bit-exactly: [x264-upstream] encoder/slicetype.c slicetype_frame_cost ->
// slicetype_slice_cost -> slicetype_mb_cost for any triple (p0, p1, b) of pictures held in the lookahead, reached from the reference at
// codec.c:1693 (x264_encoder_encode -> x264_slicetype_decide / _analyse: scenecut, --b-adapt, x264_rc_analyse_slice).
//
//   k_st_lowres : the four half-resolution phase planes of the mod-16 expanded luma with their replicated borders (frame_init_lowres_core).
//   k_st_intra  : the intra cost of every 8x8 half-resolution block on SOURCE neighbours (8x8c DC / H / V / P, filtered Intra_8x8 modes 3..8),
//                 one wavefront per 2x2 group of blocks; computed once per picture (b_intra_calculated).
//   k_st_cost   : the inter costs.  x264 walks the blocks in REVERSE raster order and predicts a block's vector from its right / lower /
//                 lower-left / lower-right neighbours of the same search, so a block depends on the row below: one wavefront per block row,
//                 rows chained bottom-up by progress counters in global memory (a 2:1 wavefront, as k_deblock2<true>); inside a block the
//                 wavefront runs the main encoder's me_search (k_mb.hip.h: 4 x 16 / 8 x 8 candidate-row lanes, reference cache in LDS) with the
//                 lookahead's settings (qp 12, me <= hex, sub-pel level 4), then the bidirectional candidates of B costs.
#include "k_mb.hip.h"
#include <new>
#include <string.h>
#include <math.h>
#include <vector>

namespace x264gpu {

constexpr int ST_MAX_B = 16, ST_MAX_SLOTS = 128;
constexpr int LOWRES_COST_MASK = (1 << 14) - 1, LOWRES_COST_SHIFT = 14;

struct StK {
    EncK ek;                                  // what me_search reads: plane geometry, the two references' plane sets, merange
    const uint8_t *i420; size_t i420_bytes; int w, h;
    int bw, bh, nb, lw, lh;
    uint8_t *cur;                             // plane set of the picture being costed / built: [S][4 planes]
    int d0, d1, dsf, bipw, do_search0, do_search1;
    int start_x, end_x, start_y, end_y;
    int param_subme, lambda;
    int16_t *mv[2]; int *mvcost[2];           // of the picture being costed, for (list, distance): [S][nb]
    const int16_t *rmv;                       // B: p1's search towards p0 (NULL: not searched yet)
    int *intra_cost; uint16_t *lowres_costs;  // [S][nb]
    int32_t *sums;                            // [S][4]: inter cost est, intra cost est, intra blocks, -
    int *progress;                            // [S][bh]
    int serial_rows;                          // one wavefront walks every row of its stream (batches) instead of one wavefront a row
};

// k_st_cost's LDS: 8x8 blocks of the half-resolution planes, me <= hex, at most two references, no chroma, no intra tiles, no record — only what me_search
// touches, so that THREE wavefronts share a SIMD where the macroblock loop's layout (20 KB) allows two (12 x 11.4 KB of the CU's 160 KB)
struct StLds {
    // the slot around an 8x8 block of a walk that runs RIGHT TO LEFT: a row of 64 columns, 44 to the block's left and 12 to its right, so that the next four
    // blocks of the row still lie inside with 12 columns to spare (the macroblock loop's 40-column slot, 12 / 20 around the block, was re-centred for nearly every
    // block: every re-centre touches ROWS x 4 cache lines whatever the row length — the L2's line rate, not its bytes, bounds this kernel); 8 rows above and below
    static constexpr int rc_mx = 44, rc_my = 8;
    using rcg = RcGeo<16, 24>;                                       // 24 rows x 64 columns of the four planes: 6 KB a slot
    __attribute__((aligned(16))) uint32_t rc[2 * RcGeo<16, 24>::SLOT_DW];        // reference-cache slots of list 0 / list 1
    uint32_t csub[CSubGeo<2>::DWORDS];
    __attribute__((aligned(16))) uint8_t src[16 * 16];
    __attribute__((aligned(16))) uint8_t csrc[8 * 16];               // (chroma rows: named by me_search's chroma-me branch, never taken here)
    uint16_t mvcost[MVC_N];
    int16_t cand[16][2];
};

__device__ __forceinline__ int st_avg4(int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; }

__global__ __launch_bounds__(256) void k_st_lowres(StK k)
{
    const int x4 = ((int)(blockIdx.x * 256 + threadIdx.x)) * 4 - PAD, y = (int)blockIdx.y - PAD, s = blockIdx.z;
    if (x4 >= k.lw + PAD) return;
    const uint8_t *src = k.i420 + (size_t)s * k.i420_bytes;
    const int cy = min(max(y, 0), k.lh - 1), Y = 2 * cy;
    const uint8_t *r0 = src + (size_t)min(Y, k.h - 1) * k.w, *r1 = src + (size_t)min(Y + 1, k.h - 1) * k.w, *r2 = src + (size_t)min(Y + 2, k.h - 1) * k.w;
    uint32_t o[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int X = 2 * min(max(x4 + i, 0), k.lw - 1);
        const int x0 = min(X, k.w - 1), x1 = min(X + 1, k.w - 1), x2 = min(X + 2, k.w - 1);
        const int a00 = r0[x0], a01 = r0[x1], a02 = r0[x2], a10 = r1[x0], a11 = r1[x1], a12 = r1[x2], a20 = r2[x0], a21 = r2[x1], a22 = r2[x2];
        o[0] |= (uint32_t)st_avg4(a00, a10, a01, a11) << (8 * i);
        o[1] |= (uint32_t)st_avg4(a01, a11, a02, a12) << (8 * i);
        o[2] |= (uint32_t)st_avg4(a10, a20, a11, a21) << (8 * i);
        o[3] |= (uint32_t)st_avg4(a11, a21, a12, a22) << (8 * i);
    }
    uint8_t *d = k.cur + (size_t)s * k.ek.luma_bytes + (size_t)(y + PAD) * k.ek.rs + x4 + PAD;
#pragma unroll
    for (int p = 0; p < 4; p++) *(uint32_t *)(d + p * k.ek.plane_bytes) = o[p];
}

// intra cost of every block + the frame's intra sums (x264: the lowres_intra_mb part of slicetype_mb_cost, run once per picture)
__global__ __launch_bounds__(256) void k_st_intra(StK k)
{
    __shared__ __attribute__((aligned(8))) uint8_t s_tab[9 * 64];
    __shared__ uint8_t s_cnb[4][4][CNB_SIZE];
    __shared__ uint8_t s_u8[4][4][U8_SIZE];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, s = blockIdx.y;
    for (int i = threadIdx.x; i < 9 * 64; i += 256) s_tab[i] = c_pred8_table[i];
    __syncthreads();
    const int gw = (k.bw + 1) >> 1, gh = (k.bh + 1) >> 1, g = blockIdx.x * 4 + wave;
    if (g >= gw * gh) return;
    const int gx = g % gw, gy = g / gw;
    const int B = lane >> 4;
    const int bx = 2 * gx + (B & 1), by = 2 * gy + (B >> 1);
    const bool bvalid = bx < k.bw && by < k.bh;
    const int bi = min(by, k.bh - 1) * k.bw + min(bx, k.bw - 1);
    const int ls = k.ek.rs;
    const uint8_t *cur0 = k.cur + (size_t)s * k.ek.luma_bytes + (size_t)PAD * ls + PAD;
    const uint8_t *blk = cur0 + (size_t)by * 8 * ls + bx * 8;
    int icost;
    {
        const int ci = (lane >> 2) & 3, j = lane & 3, t = lane & 15;
        uint8_t *cnb = s_cnb[wave][B];
        if (t < 9) cnb[CNB_TOP - 1 + t] = blk[-(long)ls - 1 + t];
        if (t < 8) cnb[CNB_LEFT + t] = blk[(long)t * ls - 1];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const PredC pp = predc_setup(cnb);
        const uint32_t e4 = *(const uint32_t *)(blk + ((ci >> 1) * 4 + j) * ls + (ci & 1) * 4);
        int best = 1 << 28;
        const int nm = k.param_subme > 1 ? 4 : 3;
        for (int m = 0; m < nm; m++) {
            const uint32_t p4 = predc_row4(cnb, pp, m, ci, j);
            best = min(best, row16_sum(k.param_subme > 1 ? satd4_half(e4, p4, lane) : (int)__builtin_amdgcn_sad_u8(e4, p4, 0u)));
        }
        if (k.param_subme > 1) {
            for (int b = 0; b < 4; b++) {
                const uint8_t *bb = cur0 + (size_t)(2 * gy + (b >> 1)) * 8 * ls + (2 * gx + (b & 1)) * 8;
                pred8_build_u(s_u8[wave][b], bb, ls, AVAIL_LEFT | AVAIL_TOP | AVAIL_TOPRIGHT | AVAIL_TOPLEFT, lane);
            }
            const int g2 = lane >> 5, b8 = (lane >> 3) & 3, row = lane & 7;
            const uint8_t *rb = cur0 + (size_t)((2 * gy + (b8 >> 1)) * 8 + row) * ls + (2 * gx + (b8 & 1)) * 8;
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
    const bool visited = bvalid && bx >= k.end_x && bx <= k.start_x && by >= k.end_y && by <= k.start_y;
    const bool score = (bx > 0 && bx < k.bw - 1 && by > 0 && by < k.bh - 1) || k.bw <= 2 || k.bh <= 2;
    if ((lane & 15) == 0 && visited) {
        k.intra_cost[(size_t)s * k.nb + bi] = icost;
        if (k.lowres_costs) k.lowres_costs[(size_t)s * k.nb + bi] = (uint16_t)min(icost, LOWRES_COST_MASK);      // lowres_costs[0][0] (I cost requests only: list_used 0)
    }
    int v = visited && score ? icost : 0, n = visited && score ? 1 : 0;
    v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
    n = __builtin_amdgcn_readlane(n, 0) + __builtin_amdgcn_readlane(n, 16) + __builtin_amdgcn_readlane(n, 32) + __builtin_amdgcn_readlane(n, 48);
    (void)n;
    if (lane == 0) atomicAdd(k.sums + (size_t)s * 4 + 1, v);
}

template <int ME>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 3))) void k_st_cost(StK k)
{
    __shared__ __attribute__((aligned(16))) StLds L;
    const int lane = threadIdx.x, s = blockIdx.y, r = lane & 15;
    // k.serial_rows: ONE wavefront walks all the rows of its stream, bottom-up (a batch of streams fills the chip by itself: no wavefront spins on the
    // row below, no pipeline to fill per stream); else one wavefront per row, chained by the progress counters (a lone stream's latency)
    // in between (the default for batches): gridDim.x wavefronts a stream, wavefront w walking rows w, w + gridDim.x, ... — the same bottom-up chain with a
    // shorter diagonal to fill and drain (a row trails the one below by two blocks: 68 wavefronts a stream idle for 136 of 256 block times at 1080p)
    const int row_first = (int)blockIdx.x, row_last = k.start_y - k.end_y, row_step = (int)gridDim.x;
    const EncK &ek = k.ek;
    {
        const uint32_t *src = (const uint32_t *)(ek.cost_all + MVCOST_HALF);
        for (int i = lane; i < MVC_N / 2; i += 64) ((uint32_t *)L.mvcost)[i] = src[i];
    }
    lds_sync();
    MeState S;
    S.mvx = S.mvy = S.mvpx = S.mvpy = S.cost = S.costmv = S.ref = S.refcost = S.cref = S.cmvx = S.cmvy = S.mvcx = S.mvcy = S.inx = S.iny = S.cdir = 0;
    WinTags wtg;
    wtg.tref = -1; wtg.tx = wtg.ty = 0;
    Prof pf;
    pf.start();
    const bool b_bidir = k.d1 > 0, satd = k.param_subme > 1;
    const int nb = k.nb, bw = k.bw, bh = k.bh, rs = ek.rs;
    int *prog = k.progress + (size_t)s * bh;
    const int row_total = k.start_x - k.end_x + 1;
    const uint8_t *cur0 = k.cur + (size_t)s * ek.luma_bytes + (size_t)PAD * rs + PAD;
    int16_t *mvl[2] = { k.mv[0] ? k.mv[0] + (size_t)s * nb * 2 : nullptr, k.mv[1] ? k.mv[1] + (size_t)s * nb * 2 : nullptr };
    int *mcl[2] = { k.mvcost[0] ? k.mvcost[0] + (size_t)s * nb : nullptr, k.mvcost[1] ? k.mvcost[1] + (size_t)s * nb : nullptr };
    const int16_t *rmv = k.rmv ? k.rmv + (size_t)s * nb * 2 : nullptr;
    const int *icosts = k.intra_cost + (size_t)s * nb;
    uint16_t *lrc = k.lowres_costs + (size_t)s * nb;
    int sum_inter = 0, sum_intra_mbs = 0;
    const s16x2 sg1 = pk_sign(lane & 1), sg2 = pk_sign(lane & 2);
    const int mvr = 2 * (ek.mv_range > 0 ? ek.mv_range : 512);

    for (int row = row_first; row <= row_last; row += row_step) {
    const int by = k.start_y - row;
    int rmv0x = 0, rmv0y = 0, rmv1x = 0, rmv1y = 0;          // this row's previous block (bx + 1) in list 0 / 1: never re-read from memory (scalars, not an array: an array
                                                           // indexed by the list lives in scratch memory)
    for (int bx = k.start_x; bx >= k.end_x; bx--) {
        if (by < k.start_y) {                              // the row below must have finished the block to the lower left
            // (no acquire fence: at agent scope that is a buffer_inv of the caches per block and, on the producer's side, a write-back of the whole L2
            //  — measured: 95 % of the wave cycles waiting whatever the occupancy.  What the row below hands over are its vectors alone: the producer
            //  writes them with agent-scope stores and waits for them before it moves its counter, this side reads them with agent-scope loads)
            const int need = min(k.start_x - (bx - 1) + 1, row_total);
            while (wfp_load<true>(prog + by + 1) < need) __builtin_amdgcn_s_sleep(2);
        }
        const int bi = by * bw + bx;
        const bool score = (bx > 0 && bx < bw - 1 && by > 0 && by < bh - 1) || bw <= 2 || bh <= 2;
        MbCtx c;
        c.s = s; c.lane = lane; c.mbx = bx; c.mby = by; c.mbi = bi; c.px = 8 * bx; c.py = 8 * by; c.sy = by;
        c.fenc = nullptr; c.fuv = nullptr; c.qp = 12; c.qpc = 12; c.lambda = k.lambda; c.subme = satd ? 4 : 2; c.satd = satd; c.chroma_me = false;
        c.cost_base = ek.cost_all; c.nref = 1;
        c.smin0 = max(4 * (-8 * bx - 12), -mvr); c.smax0 = min(4 * (8 * (bw - bx - 1) + 12), mvr - 1);
        c.smin1 = max(4 * (-8 * by - 12), -mvr); c.smax1 = min(4 * (8 * (bh - by - 1) + 12), mvr - 1);
        c.fmin0 = c.smin0 >> 2; c.fmax0 = c.smax0 >> 2; c.fmin1 = c.smin1 >> 2; c.fmax1 = c.smax1 >> 2;
        c.mvmin0 = c.smin0; c.mvmax0 = c.smax0; c.mvmin1 = c.smin1; c.mvmax1 = c.smax1;
        // the source block: rows 0..7 of the LDS source tile (me_search reads them there), and this lane's row in registers
        const uint8_t *blk = cur0 + (size_t)by * 8 * rs + bx * 8;
        lds_sync();
        uint32_t e[4] = { 0, 0, 0, 0 };
        if (r < 8) { const uint2 v = *(const uint2 *)(blk + (size_t)r * rs); e[0] = v.x; e[1] = v.y; if (lane < 8) *(uint2 *)(L.src + r * 16) = v; }
        lds_sync();
        // mbcmp of this lane group's prediction rows against the source block
        auto cmp = [&](const uint32_t p[4]) {
            if (satd) return row16_sum(satd16x4_half_pk<2>(e, p, sg1, sg2));
            unsigned sd = __builtin_amdgcn_sad_u8(p[0], e[0], 0u);
            sd = __builtin_amdgcn_sad_u8(p[1], e[1], sd);
            return row16_sum(r < 8 ? (int)sd : 0);
        };
        // this lane group's rows of the block predicted at a quarter-sample vector: from the list's reference-cache slot when it holds them (the searches
        // leave it around one of the last blocks of the row: the probe at the zero vector, the bidirectional candidates and the final pair mostly lie
        // inside), else from memory — one latency of ~2 us a probe that the wavefront cannot hide
        auto fetch = [&](int ref, int qx, int qy, uint32_t p[4]) {
            p[0] = p[1] = p[2] = p[3] = 0;
            const int X0 = rl(wtg.tx, ref), Y0 = rl(wtg.ty, ref), x0 = c.px + (qx >> 2), y0 = c.py + (qy >> 2);
            const bool inside = !ek.wp_any && rl(wtg.tref, ref) == ref && x0 >= X0 && x0 + 9 <= X0 + StLds::rcg::COLS && y0 >= Y0 && y0 + 9 <= Y0 + StLds::rcg::ROWS;
            if (inside) { if (r < 8) rc_row<StLds::rcg>(L.rc + ref * StLds::rcg::SLOT_DW, X0, Y0, c.px, c.py + r, qx, qy, false, p); }
            else if (r < 8) mc_row_global(ref_plane00(ek, s, ref), ek.plane_bytes, rs, c.px, c.py + r, qx, qy, false, p);
        };
        int i_bcost = MB_COST_MAX, list_used = 0;
        auto try_bidir = [&](int x0, int y0, int x1, int y1, int penalty) {
            uint32_t p0[4], p1[4], a[4];
            fetch(0, x0, y0, p0); fetch(1, x1, y1, p1);
            a[0] = avg_weight4_u8(p0[0], p1[0], k.bipw); a[1] = avg_weight4_u8(p0[1], p1[1], k.bipw); a[2] = a[3] = 0;
            if (r >= 8) a[0] = a[1] = 0;
            const int cst = penalty * k.lambda + uni(cmp(a));
            if (cst < i_bcost) { i_bcost = cst; list_used = 3; }
        };
        if (b_bidir) {
            int d00 = 0, d01 = 0, d10 = 0, d11 = 0;
            if (rmv) {
                const int rx = rmv[2 * bi], ry = rmv[2 * bi + 1];
                d00 = (rx * k.dsf + 128) >> 8; d01 = (ry * k.dsf + 128) >> 8;
                d10 = d00 - rx; d11 = d01 - ry;
                d00 = clampi(d00, c.smin0, c.smax0); d01 = clampi(d01, c.smin1, c.smax1);
                d10 = clampi(d10, c.smin0, c.smax0); d11 = clampi(d11, c.smin1, c.smax1);
                if (!satd) { d00 &= ~1; d01 &= ~1; d10 &= ~1; d11 &= ~1; }
                d00 = uni(d00); d01 = uni(d01); d10 = uni(d10); d11 = uni(d11);
            }
            try_bidir(d00, d01, d10, d11, 0);
            if (d00 | d01 | d10 | d11) try_bidir(0, 0, 0, 0, 0);
        }
        int mm0x = 0, mm0y = 0, mm1x = 0, mm1y = 0;          // the lists' vectors of this block
#pragma unroll
        for (int l = 0; l < 2; l++) {
            if (l && !b_bidir) break;
            int mcost;
            int16_t *mvl_l = l ? mvl[1] : mvl[0];
            int *mcl_l = l ? mcl[1] : mcl[0];
            if (l ? k.do_search1 : k.do_search0) {
                // reverse-order MV prediction: right, lower, lower-left, lower-right — candidate i in lane i of cxv / cyv
                int cxv = 0, cyv = 0, n = 0;
                auto push = [&](int x, int y) { cxv = lane == n ? x : cxv; cyv = lane == n ? y : cyv; n++; };
                if (bx < bw - 1) push(l ? rmv1x : rmv0x, l ? rmv1y : rmv0y);
                if (by < bh - 1) {
                    const int16_t *lo = mvl_l + 2 * (bi + bw);
                    int vx = 0, vy = 0;
                    if (lane < 3) {
                        const int o = lane == 0 ? 0 : lane == 1 ? -1 : 1;
                        if ((o < 0 && bx > 0) || o == 0 || (o > 0 && bx < bw - 1)) {
                            const int w = __hip_atomic_load((const int *)(lo + 2 * o), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (another wavefront's, of this launch)
                            vx = (int)(int16_t)(w & 0xffff); vy = (int)(int16_t)(w >> 16);
                        }
                    }
                    push(rl(vx, 0), rl(vy, 0));
                    if (bx > 0) push(rl(vx, 1), rl(vy, 1));
                    if (bx < bw - 1) push(rl(vx, 2), rl(vy, 2));
                }
                int mvpx, mvpy;
                if (n <= 1) { mvpx = rl(cxv, 0); mvpy = rl(cyv, 0); }
                else { mvpx = median3(rl(cxv, 0), rl(cxv, 1), rl(cxv, 2)); mvpy = median3(rl(cyv, 0), rl(cyv, 1), rl(cyv, 2)); }
                mvpx = uni(mvpx); mvpy = uni(mvpy);
                bool skip = false;
                int mx = 0, my = 0;
                mcost = 0;
                if (!(mvpx | mvpy)) {
                    uint32_t p[4];
                    fetch(l, 0, 0, p);
                    mcost = uni(cmp(p));
                    skip = mcost < 64;
                }
                if (!skip) {
                    S.inx = cxv; S.iny = cyv;
                    MeJob jb;
                    jb.W = 8; jb.H = 8; jb.ox = 0; jb.oy = 0; jb.ref = l; jb.mvpx = mvpx; jb.mvpy = mvpy; jb.n_mvc = n; jb.search = true; jb.qonly = false;
                    jb.hp_it = 1; jb.qp_it = satd ? 1 : 0; jb.use_thresh = false;
                    int cost = 0, cost_mv = 0, thresh = 0x7fffffff;
                    lds_sync();
                    me_search<2, ME, true, StLds>(ek, L, c, jb, mx, my, cost, cost_mv, thresh, S, wtg, pf);
                    lds_sync();
                    mcost = cost - (int)ek.cost_all[MVCOST_HALF];         // "remove mvcost from skip mbs"
                    if (mx | my) mcost += 5 * k.lambda;
                }
                if (l) { mm1x = mx; mm1y = my; } else { mm0x = mx; mm0y = my; }
                if (lane == 0) { __hip_atomic_store((int *)(mvl_l + 2 * bi), (int)(((unsigned)my << 16) | ((unsigned)mx & 0xffffu)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); mcl_l[bi] = mcost; }
            } else {
                const int mx = uni((int)mvl_l[2 * bi]), my = uni((int)mvl_l[2 * bi + 1]); mcost = uni(mcl_l[bi]);
                if (l) { mm1x = mx; mm1y = my; } else { mm0x = mx; mm0y = my; }
            }
            if (l) { rmv1x = mm1x; rmv1y = mm1y; } else { rmv0x = mm0x; rmv0y = mm0y; }
            if (mcost < i_bcost) { i_bcost = mcost; list_used = l + 1; }
        }
        if (b_bidir && (mm0x | mm0y | mm1x | mm1y)) try_bidir(mm0x, mm0y, mm1x, mm1y, 5);
        i_bcost += 4;                                      // lowres_penalty
        if (!b_bidir) {                                    // intra blocks are not considered in B pictures
            const int icost = uni(icosts[bi]);
            if (icost < i_bcost) { i_bcost = icost; list_used = 0; if (score) sum_intra_mbs++; }
        }
        if (score) sum_inter += i_bcost;
        if (lane == 0) lrc[bi] = (uint16_t)(min(i_bcost, LOWRES_COST_MASK) + (list_used << LOWRES_COST_SHIFT));
        __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): the vectors have arrived where the row above reads them, before the counter says so
        if (lane == 0) wfp_store<true>(prog + by, k.start_x - bx + 1);
    }
    }
    if (lane == 0) { atomicAdd(k.sums + (size_t)s * 4, sum_inter); atomicAdd(k.sums + (size_t)s * 4 + 2, sum_intra_mbs); }
}

// ---- the primitives behind x264_weights_analyse (oracle/slicetype.c x264o_slicetype_pixel_stats / _weight_cost) ----
struct StWeightK {
    int bw, bh, nb, rs, satd, wpk;               // wpk: EncK::wl0 packing of the explicit luma weight, 0 = unweighted
    size_t plane_bytes, luma_bytes, i420_bytes; int w, h, cw, ch;
    const uint8_t *i420; const uint8_t *fenc, *ref; const int16_t *mv; const int *intra_cost;
    unsigned long long *out;                     // [S][2]
};
__global__ __launch_bounds__(256) void k_st_pixel_stats(StWeightK k)
{
    const int s = blockIdx.y;
    const uint8_t *src = k.i420 + (size_t)s * k.i420_bytes;
    unsigned long long sum = 0, sqr = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < k.cw * k.ch; i += gridDim.x * 256) {
        const int y = i / k.cw, x = i - y * k.cw;
        const unsigned p = src[(size_t)min(y, k.h - 1) * k.w + min(x, k.w - 1)];
        sum += p; sqr += p * p;
    }
    for (int o = 32; o; o >>= 1) { sum += __shfl_xor(sum, o); sqr += __shfl_xor(sqr, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(k.out + (size_t)s * 2, sum); atomicAdd(k.out + (size_t)s * 2 + 1, sqr); }
}
// weight_cost_luma: lane = (block of the group of eight, row); every block of the picture
__global__ __launch_bounds__(64) void k_st_weight_cost(StWeightK k)
{
    const int lane = threadIdx.x, s = blockIdx.y, row = lane & 7;
    const int bi = blockIdx.x * 8 + (lane >> 3);
    const bool valid = bi < k.nb;
    const int b = valid ? bi : 0, bx = b % k.bw, by = b / k.bw;
    const uint8_t *f0 = k.fenc + (size_t)s * k.luma_bytes + (size_t)PAD * k.rs + PAD;
    const uint8_t *r0 = k.ref + (size_t)s * k.luma_bytes + (size_t)PAD * k.rs + PAD;
    int mvx = 0, mvy = 0;
    if (k.mv) { const int16_t *m = k.mv + ((size_t)s * k.nb + b) * 2; mvx = m[0]; mvy = m[1]; }
    const uint2 e = *(const uint2 *)(f0 + (size_t)(by * 8 + row) * k.rs + bx * 8);
    uint32_t p[4];
    mc_row_global(r0, k.plane_bytes, k.rs, bx * 8, by * 8 + row, mvx, mvy, false, p);
    if (k.wpk) { p[0] = wp4(p[0], k.wpk); p[1] = wp4(p[1], k.wpk); }
    int h;
    if (k.satd) { h = satd4_half(e.x, p[0], lane) + satd4_half(e.y, p[1], lane); h = quad_sum(h); h += xor4(h); }
    else { h = (int)__builtin_amdgcn_sad_u8(p[1], e.y, __builtin_amdgcn_sad_u8(p[0], e.x, 0u)); h = quad_sum(h); h += xor4(h); }
    const int cost = min(h, k.intra_cost[(size_t)s * k.nb + b]);
    unsigned long long v = valid && row == 0 ? (unsigned long long)cost : 0ull;
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) atomicAdd(k.out + (size_t)s * 2, v);
}

// i_pixel_sum / i_pixel_ssd of the chroma planes (blockIdx.z = plane Cb / Cr): raw sums, the caller turns the squares into the ssd
struct StChromaK {
    int w, h, pw, ph, cw, ch, bw, nb, plane, on, scale, denom, offset;
    size_t i420_bytes;
    const uint8_t *i420, *ref; const int16_t *mv;
    unsigned long long *out;                     // stats: [S][4]; cost: [S][2]
};
__global__ __launch_bounds__(256) void k_st_chroma_stats(StChromaK k)
{
    const int s = blockIdx.y, c = blockIdx.z;
    const uint8_t *src = k.i420 + (size_t)s * k.i420_bytes + (size_t)k.w * k.h + (size_t)c * k.pw * k.ph;
    unsigned long long sum = 0, sqr = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < k.cw * k.ch; i += gridDim.x * 256) {
        const int y = i / k.cw, x = i - y * k.cw;
        const unsigned p = src[(size_t)min(y, k.ph - 1) * k.pw + min(x, k.pw - 1)];
        sum += p; sqr += p * p;
    }
    for (int o = 32; o; o >>= 1) { sum += __shfl_xor(sum, o); sqr += __shfl_xor(sqr, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(k.out + (size_t)s * 4 + 2 * c, sum); atomicAdd(k.out + (size_t)s * 4 + 2 * c + 1, sqr); }
}
// weight_cost_chroma (oracle x264o_slicetype_weight_cost_chroma): one lane per sample of an 8x8 chroma block — mc_chroma with the block's
// half-resolution vector, the weight, the difference to the source; |sum| per block (pixf.asd8).  Picture edges by clamping (the expanded border)
__global__ __launch_bounds__(64) void k_st_weight_cost_chroma(StChromaK k)
{
    const int lane = threadIdx.x, s = blockIdx.y, b = blockIdx.x, bx = b % k.bw, by = b / k.bw, x = lane & 7, y = lane >> 3;
    const uint8_t *src = k.i420 + (size_t)s * k.i420_bytes + (size_t)k.w * k.h + (size_t)(k.plane - 1) * k.pw * k.ph;
    const uint8_t *ref = k.ref + (size_t)s * k.i420_bytes + (size_t)k.w * k.h + (size_t)(k.plane - 1) * k.pw * k.ph;
    int mvx = 0, mvy = 0;
    if (k.mv) { const int16_t *m = k.mv + ((size_t)s * k.nb + b) * 2; mvx = m[0]; mvy = m[1]; }
    const int d8x = mvx & 7, d8y = mvy & 7, cA = (8 - d8x) * (8 - d8y), cB = d8x * (8 - d8y), cC = (8 - d8x) * d8y, cD = d8x * d8y;
    const int px = bx * 8 + x + (mvx >> 3), py = by * 8 + y + (mvy >> 3);
    const int x0 = min(max(px, 0), k.pw - 1), x1 = min(max(px + 1, 0), k.pw - 1), y0 = min(max(py, 0), k.ph - 1), y1 = min(max(py + 1, 0), k.ph - 1);
    int p = (cA * ref[(size_t)y0 * k.pw + x0] + cB * ref[(size_t)y0 * k.pw + x1] + cC * ref[(size_t)y1 * k.pw + x0] + cD * ref[(size_t)y1 * k.pw + x1] + 32) >> 6;
    if (k.on) { p = k.denom >= 1 ? ((p * k.scale + (1 << (k.denom - 1))) >> k.denom) + k.offset : p * k.scale + k.offset; p = min(max(p, 0), 255); }
    int d = p - (int)src[(size_t)min(by * 8 + y, k.ph - 1) * k.pw + min(bx * 8 + x, k.pw - 1)];
    for (int o = 32; o; o >>= 1) d += __shfl_xor(d, o);
    if (lane == 0) atomicAdd(k.out + (size_t)s * 2, (unsigned long long)abs(d));
}

// ---- macroblock-tree through B pictures (oracle/slicetype.c x264o_slicetype_propagate / _finish; x264 macroblock_tree_propagate,
//      mbtree_propagate_cost / _list, macroblock_tree_finish).  Sums saturate at 32767 in x264; every addend is non-negative, so the
//      accumulators here are plain 32-bit atomics and the saturation is applied where a sum is READ ----
__device__ __forceinline__ int st_inv_qscale(float aq) { return x264_exp2fix8(aq); }          // frame->i_inv_qscale_factor
struct StTreeK {
    int bw, bh, nb, d0, d1, bipw0, bipw1, referenced; float strength, weightdelta;
    const int *intra_cost; const uint16_t *lowres_costs; const int16_t *mv[2]; const float *aq;
    const int32_t *prop_b; int32_t *prop_ref[2];
    float *out;
};
__global__ __launch_bounds__(256) void k_st_propagate(StTreeK k)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= k.nb) return;
    const size_t o = (size_t)s * k.nb + i;
    const int bx = i % k.bw, by = i / k.bw;
    const int intra = min(k.intra_cost[o], LOWRES_COST_MASK), lc = k.lowres_costs[o];
    const int best = lc & LOWRES_COST_MASK, inter = min(best, intra), inv = st_inv_qscale(k.aq ? k.aq[o] : 0.f);
    const int amount = x264_propagate_amount(k.referenced ? min(k.prop_b[o], 32767) : 0, intra, inter, inv);          // mbtree_propagate_cost
    const int used = lc >> LOWRES_COST_SHIFT;
    for (int l = 0; l < (k.d1 > 0 ? 2 : 1); l++) {
        if (!(used & (1 << l))) continue;
        int la = amount;
        if (used == 3) la = (la * (l ? k.bipw1 : k.bipw0) + 32) >> 6;
        const int16_t *mv = k.mv[l] + o * 2;
        int x = mv[0], y = mv[1];
        int32_t *ref = k.prop_ref[l] + (size_t)s * k.nb;
        if (!(x | y)) { atomicAdd(ref + i, la); continue; }
        const int mbx = (x >> 5) + bx, mby = (y >> 5) + by;
        x &= 31; y &= 31;
        const int w0 = ((32 - y) * (32 - x) * la + 512) >> 10, w1 = ((32 - y) * x * la + 512) >> 10;
        const int w2 = (y * (32 - x) * la + 512) >> 10, w3 = (y * x * la + 512) >> 10;
        if (mby >= 0 && mby < k.bh) { if (mbx >= 0 && mbx < k.bw) atomicAdd(ref + mby * k.bw + mbx, w0); if (mbx + 1 >= 0 && mbx + 1 < k.bw) atomicAdd(ref + mby * k.bw + mbx + 1, w1); }
        if (mby + 1 >= 0 && mby + 1 < k.bh) { if (mbx >= 0 && mbx < k.bw) atomicAdd(ref + (mby + 1) * k.bw + mbx, w2); if (mbx + 1 >= 0 && mbx + 1 < k.bw) atomicAdd(ref + (mby + 1) * k.bw + mbx + 1, w3); }
    }
}
__global__ __launch_bounds__(256) void k_st_finish(StTreeK k)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i >= k.nb) return;
    const size_t o = (size_t)s * k.nb + i;
    const float a = k.aq ? k.aq[o] : 0.f;
    const int icost = min(k.intra_cost[o], LOWRES_COST_MASK);
    const int intra = (icost * st_inv_qscale(a) + 128) >> 8;
    float off = a;
    if (intra) {
        const int p2 = min(k.prop_b[o], 32767) * 2;          // (propagate * fps_factor + 128) >> 8, fps_factor = 512
        const float log2_ratio = f_add(f_sub(x264_log2((unsigned)(intra + p2)), x264_log2((unsigned)intra)), k.weightdelta);
        off = f_sub(a, f_mul(k.strength, log2_ratio));
    }
    k.out[o] = off;
}

}  // namespace x264gpu

using namespace x264gpu;

struct x264gpu_slicetype {
    int w, h, streams, slots, bframes, bw, bh, nb, lw, lh, ls;
    int me_method, param_subme, me_range, weightb, mv_range, do_edges, lambda, bframe_bias;
    size_t lplane, lpic;
    uint8_t *planes[ST_MAX_SLOTS];
    int16_t *mvs[ST_MAX_SLOTS][2][ST_MAX_B + 1]; int *mvcosts[ST_MAX_SLOTS][2][ST_MAX_B + 1];
    bool searched[ST_MAX_SLOTS][2][ST_MAX_B + 1];
    int *intra_cost[ST_MAX_SLOTS]; bool intra_calculated[ST_MAX_SLOTS];
    uint16_t *lowres_costs[ST_MAX_SLOTS];     // [(d0 * (bframes + 2) + d1)][S][nb]
    std::vector<int32_t> cost_est[ST_MAX_SLOTS];       // [(d0 * (bframes + 2) + d1) * S + s], -1 = not computed
    std::vector<int32_t> intra_mbs[ST_MAX_SLOTS];      // [d0 * S + s]
    uint16_t *cost_mv; int32_t *sums; int *progress; int serial_rows = -1;          // serial_rows: -1 auto (= 0: measured faster at every batch size), 0 / 1 forced
    std::vector<int32_t> h_sums;
    int32_t *prop[ST_MAX_SLOTS]; float *aq[ST_MAX_SLOTS]; bool have_aq[ST_MAX_SLOTS];        // macroblock-tree: propagate costs, AQ offsets (f_qp_offset_aq)
    unsigned long long *d_acc;                                                                // [S][4] accumulators of the weight primitives
    std::vector<unsigned long long> cstats[ST_MAX_SLOTS];                                     // per slot [S][4]: i_pixel_sum, i_pixel_ssd of Cb, of Cr
    std::vector<unsigned long long> stats[ST_MAX_SLOTS];                                      // per slot [S][2]: i_pixel_sum, i_pixel_ssd of the luma
};

extern "C" {

void x264gpu_slicetype_destroy(x264gpu_slicetype *st)
{
    if (!st) return;
    for (int i = 0; i < st->slots; i++) {
        (void)hipFree(st->planes[i]); (void)hipFree(st->intra_cost[i]); (void)hipFree(st->lowres_costs[i]); (void)hipFree(st->prop[i]); (void)hipFree(st->aq[i]);
        for (int l = 0; l < 2; l++) for (int d = 0; d <= st->bframes; d++) { (void)hipFree(st->mvs[i][l][d]); (void)hipFree(st->mvcosts[i][l][d]); }
    }
    (void)hipFree(st->cost_mv); (void)hipFree(st->sums); (void)hipFree(st->progress); (void)hipFree(st->d_acc);
    delete st;
}

int x264gpu_slicetype_create(x264gpu_slicetype **out, int width, int height, int streams, int slots, int bframes, int me_method, int subme, int me_range,
                             int weightb, int mv_range, int do_edges)
{
    ARG_TRY(out && width >= 16 && height >= 16 && !(width & 1) && !(height & 1) && width <= 4096 && height <= 2304 && streams >= 1);
    ARG_TRY(slots >= 2 && slots <= ST_MAX_SLOTS && bframes >= 0 && bframes <= ST_MAX_B);
    x264gpu_slicetype *st = new (std::nothrow) x264gpu_slicetype();
    if (!st) return set_err(X264GPU_ENOMEM, "slicetype", hipSuccess);
    st->w = width; st->h = height; st->streams = streams; st->slots = slots; st->bframes = bframes;
    st->bw = (width + 15) / 16; st->bh = (height + 15) / 16; st->nb = st->bw * st->bh;
    st->lw = st->bw * 8; st->lh = st->bh * 8;
    st->ls = (st->lw + 2 * PAD + 63) / 64 * 64;
    st->lplane = (size_t)st->ls * (st->lh + 2 * PAD); st->lpic = 4 * st->lplane;
    st->me_method = subme > 1 ? (me_method < 1 ? me_method : 1) : 0;         // min(hex, --me), or dia (lowres_context_init)
    st->param_subme = subme; st->me_range = me_range < 4 ? 4 : me_range > 16 ? 16 : me_range;
    st->weightb = weightb; st->mv_range = mv_range > 0 ? mv_range : 512;
    st->do_edges = do_edges || st->bw <= 2 || st->bh <= 2;
    st->lambda = lambda_of(12);                                              // X264_LOOKAHEAD_QP
    const size_t S = streams, nb = st->nb, nd = (size_t)(bframes + 2) * (bframes + 2);
    hipError_t er = hipSuccess;
    auto alloc = [&](void **p, size_t n) { if (er == hipSuccess) er = hipMalloc(p, n); if (er == hipSuccess) er = hipMemset(*p, 0, n); };
    for (int i = 0; i < slots; i++) {
        alloc((void **)&st->planes[i], S * st->lpic);
        alloc((void **)&st->intra_cost[i], S * nb * sizeof(int));
        alloc((void **)&st->prop[i], S * nb * sizeof(int32_t)); alloc((void **)&st->aq[i], S * nb * sizeof(float));
        alloc((void **)&st->lowres_costs[i], nd * S * nb * sizeof(uint16_t));
        for (int l = 0; l < 2; l++) for (int d = 0; d <= bframes; d++) { alloc((void **)&st->mvs[i][l][d], S * nb * 2 * sizeof(int16_t)); alloc((void **)&st->mvcosts[i][l][d], S * nb * sizeof(int)); }
        st->cost_est[i].assign(nd * S, -1); st->intra_mbs[i].assign((size_t)(bframes + 2) * S, 0);
    }
    alloc((void **)&st->sums, S * 4 * sizeof(int32_t));
    alloc((void **)&st->d_acc, S * 4 * sizeof(unsigned long long));
    alloc((void **)&st->progress, S * (size_t)st->bh * sizeof(int));
    alloc((void **)&st->cost_mv, 2 * MVCOST_HALF * sizeof(uint16_t));
    st->h_sums.resize(S * 4);
    if (er == hipSuccess) {
        std::vector<uint16_t> hc(2 * MVCOST_HALF);
        for (int i = 0; i < MVCOST_HALF; i++) {
            const float bits = log2f((float)(i + 1)) * 2.0f + 0.718f + (i ? 1.0f : 0.0f);
            int c = (int)((float)st->lambda * bits + 0.5f);
            if (c > 65535) c = 65535;
            hc[MVCOST_HALF + i] = (uint16_t)c; hc[MVCOST_HALF - i] = (uint16_t)c;
        }
        hc[0] = hc[1];
        er = hipMemcpy(st->cost_mv, hc.data(), hc.size() * sizeof(uint16_t), hipMemcpyHostToDevice);
    }
    if (er != hipSuccess) { x264gpu_slicetype_destroy(st); return set_err(er == hipErrorOutOfMemory ? X264GPU_ENOMEM : X264GPU_EHIP, "slicetype buffers", er); }
    *out = st;
    return X264GPU_OK;
}

static void st_fill(const x264gpu_slicetype *st, StK &k)
{
    memset(&k, 0, sizeof(k));
    k.w = st->w; k.h = st->h; k.i420_bytes = (size_t)st->w * st->h * 3 / 2;
    k.bw = st->bw; k.bh = st->bh; k.nb = st->nb; k.lw = st->lw; k.lh = st->lh;
    k.param_subme = st->param_subme; k.lambda = st->lambda;
    k.start_y = st->bh - 2 + st->do_edges; k.end_y = 1 - st->do_edges; k.start_x = st->bw - 2 + st->do_edges; k.end_x = 1 - st->do_edges;
    k.sums = st->sums; k.progress = st->progress;
    EncK &e = k.ek;
    e.w = st->lw; e.h = st->lh; e.cw = st->lw; e.ch = st->lh; e.rs = st->ls; e.plane_bytes = st->lplane; e.luma_bytes = st->lpic;
    e.me_range = st->me_range; e.me_method = st->me_method; e.subme = st->param_subme > 1 ? 4 : 2; e.mv_range = st->mv_range;
    e.cost_all = st->cost_mv; e.nref = 1; e.slices = 1;
}

int x264gpu_slicetype_put_frame(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, void *stream)
{
    ARG_TRY(st && d_i420 && slot >= 0 && slot < st->slots);
    hipStream_t s = (hipStream_t)stream;
    StK k;
    st_fill(st, k);
    k.i420 = d_i420; k.cur = st->planes[slot];
    hipLaunchKernelGGL(k_st_lowres, dim3(((st->lw + 2 * PAD) / 4 + 255) / 256, st->lh + 2 * PAD, st->streams), dim3(256), 0, s, k);
    const size_t S = st->streams, nb = st->nb;
    for (int l = 0; l < 2; l++)
        for (int d = 0; d <= st->bframes; d++) {
            st->searched[slot][l][d] = false;
            HIP_TRY(hipMemsetAsync(st->mvs[slot][l][d], 0, S * nb * 2 * sizeof(int16_t), s));
        }
    st->intra_calculated[slot] = false;
    st->have_aq[slot] = false;
    st->stats[slot].clear(); st->cstats[slot].clear();
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)st->intra_cost[slot], 0xffff, S * nb, s));          // x264_frame_new: memset( i_intra_cost, -1 )
    HIP_TRY(hipMemsetAsync(st->prop[slot], 0, S * nb * sizeof(int32_t), s));
    std::fill(st->cost_est[slot].begin(), st->cost_est[slot].end(), -1);
    std::fill(st->intra_mbs[slot].begin(), st->intra_mbs[slot].end(), 0);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

// slicetype_frame_cost(p0, p1, b): the slots of the three pictures and the distances d0 = b - p0, d1 = p1 - b; h_score[streams] receives the scores
// (the call synchronises on the stream when it has to compute)
int x264gpu_slicetype_frame_cost_w(x264gpu_slicetype *st, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset, int32_t *h_score, void *stream);
int x264gpu_slicetype_frame_cost(x264gpu_slicetype *st, int s0, int s1, int sb, int d0, int d1, int32_t *h_score, void *stream)
{
    return x264gpu_slicetype_frame_cost_w(st, s0, s1, sb, d0, d1, 0, 1, 0, 0, h_score, stream);
}
// ... with the explicit luma weight x264_weights_analyse( b_lookahead = 1 ) found for a P cost whose search has not run yet: the list-0 search runs on
// the weighted reference
int x264gpu_slicetype_frame_cost_w(x264gpu_slicetype *st, int s0, int s1, int sb, int d0, int d1, int on, int scale, int denom, int offset, int32_t *h_score, void *stream)
{
    ARG_TRY(!on || (denom >= 0 && denom <= 7 && scale >= -128 && scale <= 127 && offset >= -128 && offset <= 127));
    ARG_TRY(st && h_score && s0 >= 0 && s0 < st->slots && s1 >= 0 && s1 < st->slots && sb >= 0 && sb < st->slots);
    ARG_TRY(d0 >= 0 && d1 >= 0 && d0 <= st->bframes + 1 && d1 <= st->bframes + 1 && !(d1 > 0 && d0 == 0));
    const size_t S = st->streams, nb = st->nb;
    const int nd = st->bframes + 2;
    int32_t *memo = st->cost_est[sb].data() + (size_t)(d0 * nd + d1) * S;
    if (memo[0] >= 0) { memcpy(h_score, memo, S * sizeof(int32_t)); return X264GPU_OK; }
    hipStream_t s = (hipStream_t)stream;
    StK k;
    st_fill(st, k);
    k.cur = st->planes[sb];
    k.intra_cost = st->intra_cost[sb];
    HIP_TRY(hipMemsetAsync(st->sums, 0, S * 4 * sizeof(int32_t), s));
    const bool need_intra = !st->intra_calculated[sb];
    if (need_intra) {
        k.lowres_costs = d0 == 0 && d1 == 0 ? st->lowres_costs[sb] : nullptr;
        const int groups = ((st->bw + 1) / 2) * ((st->bh + 1) / 2);
        hipLaunchKernelGGL(k_st_intra, dim3((groups + 3) / 4, st->streams), dim3(256), 0, s, k);
    }
    const bool is_i = d0 == 0 && d1 == 0;
    if (!is_i) {
        k.d0 = d0; k.d1 = d1;
        k.dsf = d1 > 0 ? ((d0 << 8) + ((d0 + d1) >> 1)) / (d0 + d1) : 128;
        k.bipw = st->weightb ? 64 - (k.dsf >> 2) : 32;
        k.do_search0 = !st->searched[sb][0][d0 - 1];
        k.do_search1 = d1 > 0 && !st->searched[sb][1][d1 - 1];
        k.mv[0] = st->mvs[sb][0][d0 - 1]; k.mvcost[0] = st->mvcosts[sb][0][d0 - 1];
        if (d1 > 0) { k.mv[1] = st->mvs[sb][1][d1 - 1]; k.mvcost[1] = st->mvcosts[sb][1][d1 - 1]; }
        k.rmv = d1 > 0 && st->searched[s1][0][d0 + d1 - 1] ? st->mvs[s1][0][d0 + d1 - 1] : nullptr;
        k.lowres_costs = st->lowres_costs[sb] + (size_t)(d0 * nd + d1) * S * nb;
        k.ek.ref_luma[0] = st->planes[s0]; k.ek.ref_luma[1] = st->planes[s1];
        if (on && d1 == 0) { k.ek.wp_any = 1; k.ek.wl0[0] = (int)(uint8_t)(int8_t)offset | (int)(uint8_t)(int8_t)scale << 8 | denom << 16 | 1 << 24; }
        for (int r = 2; r < 8; r++) k.ek.ref_luma[r] = st->planes[s0];
        HIP_TRY(hipMemsetAsync(st->progress, 0, S * (size_t)st->bh * sizeof(int), s));
        const int rows = k.start_y - k.end_y + 1;
        if (rows > 0 && k.start_x >= k.end_x) {
            // one wavefront a row, chained bottom-up (default), or — x264gpu_slicetype_set_row_mode(1) — one wavefront a stream walking its rows itself.
            // Measured at 2048 streams of 1080p (bench.py `lookahead`): the single wavefront is 8 % SLOWER (17.8 s vs 16.5 s for 23 costs): the rows'
            // pipeline was not what the costs wait for, the searches are; so auto = the row pipeline at every batch size
            // auto (round 6): as many wavefronts a stream as fill the chip about four times over — 2048 streams: 6 (the diagonal of 68 wavefronts a stream
            // spent half its time filling and draining, polling the row below from memory all the while), one stream: a wavefront a row (its latency)
            k.serial_rows = st->serial_rows < 0 ? 0 : st->serial_rows;
            const int fill = (4 * 3072 + st->streams - 1) / st->streams;
            const int gx = k.serial_rows ? 1 : st->serial_rows == 0 ? rows : fill < 1 ? 1 : fill < rows ? fill : rows;
            if (st->me_method == 0) hipLaunchKernelGGL(k_st_cost<0>, dim3(gx, st->streams), dim3(64), 0, s, k);
            else hipLaunchKernelGGL(k_st_cost<1>, dim3(gx, st->streams), dim3(64), 0, s, k);
        }
        st->searched[sb][0][d0 - 1] = true;
        if (d1 > 0) st->searched[sb][1][d1 - 1] = true;
    }
    HIP_TRY(hipMemcpyAsync(st->h_sums.data(), st->sums, S * 4 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipGetLastError());
    int32_t *memo_i = st->cost_est[sb].data();                 // [0][0]
    for (size_t i = 0; i < S; i++) {
        const int32_t *v = st->h_sums.data() + 4 * i;
        if (need_intra) memo_i[i] = v[1];
        if (d1 == 0) st->intra_mbs[sb][(size_t)d0 * S + i] = v[2];
        int64_t score = is_i ? memo_i[i] : v[0];
        if (d1 > 0) score = score * 100 / (120 + st->bframe_bias);
        memo[i] = (int32_t)score;
    }
    st->intra_calculated[sb] = true;
    memcpy(h_score, memo, S * sizeof(int32_t));
    return X264GPU_OK;
}

// ---- x264_weights_analyse's primitives ----
// i_pixel_sum / i_pixel_ssd of the luma of the picture last put into `slot` (d_i420: that picture again): h_out[streams][2]
int x264gpu_slicetype_pixel_stats(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, uint64_t *h_out, void *stream)
{
    ARG_TRY(st && d_i420 && h_out && slot >= 0 && slot < st->slots);
    const size_t S = st->streams;
    if (st->stats[slot].empty()) {
        hipStream_t s = (hipStream_t)stream;
        StWeightK k;
        memset(&k, 0, sizeof(k));
        k.w = st->w; k.h = st->h; k.cw = st->bw * 16; k.ch = st->bh * 16; k.i420 = d_i420; k.i420_bytes = (size_t)st->w * st->h * 3 / 2; k.out = st->d_acc;
        HIP_TRY(hipMemsetAsync(st->d_acc, 0, S * 2 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_st_pixel_stats, dim3(64, st->streams), dim3(256), 0, s, k);
        st->stats[slot].resize(S * 2);
        HIP_TRY(hipMemcpyAsync(st->stats[slot].data(), st->d_acc, S * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const unsigned long long n = (unsigned long long)k.cw * k.ch;
        for (size_t i = 0; i < S; i++) { const unsigned long long sum = st->stats[slot][2 * i], sqr = st->stats[slot][2 * i + 1]; st->stats[slot][2 * i + 1] = sqr - (sum * sum + n / 2) / n; }
    }
    memcpy(h_out, st->stats[slot].data(), S * 2 * sizeof(uint64_t));
    return X264GPU_OK;
}
// weight_cost_luma of the picture in slot_fenc against the one in slot_ref under the explicit luma weight (on = 0: none); dist > 0: the reference
// is motion-compensated by the picture's list-0 vectors of that distance when that search has run.  h_cost[streams]
int x264gpu_slicetype_weight_cost(x264gpu_slicetype *st, int slot_fenc, int slot_ref, int dist, int on, int scale, int denom, int offset, int64_t *h_cost, void *stream)
{
    ARG_TRY(st && h_cost && slot_fenc >= 0 && slot_fenc < st->slots && slot_ref >= 0 && slot_ref < st->slots && st->intra_calculated[slot_fenc]);
    ARG_TRY(!on || (denom >= 0 && denom <= 7 && scale >= -128 && scale <= 127 && offset >= -128 && offset <= 127));
    hipStream_t s = (hipStream_t)stream;
    const size_t S = st->streams;
    StWeightK k;
    memset(&k, 0, sizeof(k));
    k.bw = st->bw; k.bh = st->bh; k.nb = st->nb; k.rs = st->ls; k.satd = st->param_subme > 1; k.plane_bytes = st->lplane; k.luma_bytes = st->lpic;
    k.wpk = on ? ((int)(uint8_t)(int8_t)offset | (int)(uint8_t)(int8_t)scale << 8 | denom << 16 | 1 << 24) : 0;
    k.fenc = st->planes[slot_fenc]; k.ref = st->planes[slot_ref];
    k.mv = dist > 0 && dist <= st->bframes + 1 && st->searched[slot_fenc][0][dist - 1] ? st->mvs[slot_fenc][0][dist - 1] : nullptr;
    k.intra_cost = st->intra_cost[slot_fenc]; k.out = st->d_acc;
    HIP_TRY(hipMemsetAsync(st->d_acc, 0, S * 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_st_weight_cost, dim3((st->nb + 7) / 8, st->streams), dim3(64), 0, s, k);
    std::vector<unsigned long long> v(S * 2);
    HIP_TRY(hipMemcpyAsync(v.data(), st->d_acc, S * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    auto size_ue = [](int x) { int n = 0; x++; while (x >> (n + 1)) n++; return 2 * n + 1; };
    const int hdr = on ? st->lambda * (10 + size_ue(denom) * 2 + 2 * (size_ue(scale > 0 ? 2 * scale - 1 : -2 * scale) + size_ue(offset > 0 ? 2 * offset - 1 : -2 * offset))) : 0;
    for (size_t i = 0; i < S; i++) h_cost[i] = (int64_t)v[2 * i] + hdr;
    return X264GPU_OK;
}

// ... of the chroma planes: h_out[streams][4] = { sum Cb, ssd Cb, sum Cr, ssd Cr }
int x264gpu_slicetype_chroma_stats(x264gpu_slicetype *st, int slot, const uint8_t *d_i420, uint64_t *h_out, void *stream)
{
    ARG_TRY(st && d_i420 && h_out && slot >= 0 && slot < st->slots);
    const size_t S = st->streams;
    if (st->cstats[slot].empty()) {
        hipStream_t s = (hipStream_t)stream;
        StChromaK k;
        memset(&k, 0, sizeof(k));
        k.w = st->w; k.h = st->h; k.pw = st->w / 2; k.ph = st->h / 2; k.cw = st->bw * 8; k.ch = st->bh * 8; k.i420 = d_i420; k.i420_bytes = (size_t)st->w * st->h * 3 / 2; k.out = st->d_acc;
        HIP_TRY(hipMemsetAsync(st->d_acc, 0, S * 4 * sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_st_chroma_stats, dim3(32, st->streams, 2), dim3(256), 0, s, k);
        st->cstats[slot].resize(S * 4);
        HIP_TRY(hipMemcpyAsync(st->cstats[slot].data(), st->d_acc, S * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const unsigned long long n = (unsigned long long)k.cw * k.ch;
        for (size_t i = 0; i < S * 2; i++) { const unsigned long long sum = st->cstats[slot][2 * i], sqr = st->cstats[slot][2 * i + 1]; st->cstats[slot][2 * i + 1] = sqr - (sum * sum + n / 2) / n; }
    }
    memcpy(h_out, st->cstats[slot].data(), S * 4 * sizeof(uint64_t));
    return X264GPU_OK;
}
// weight_cost_chroma of plane 1 (Cb) / 2 (Cr) of the raw picture d_i420_fenc (the one in slot_fenc) against the raw picture d_i420_ref.  h_cost[streams]
int x264gpu_slicetype_weight_cost_chroma(x264gpu_slicetype *st, int slot_fenc, const uint8_t *d_i420_fenc, const uint8_t *d_i420_ref, int dist, int plane,
                                         int on, int scale, int denom, int offset, int64_t *h_cost, void *stream)
{
    ARG_TRY(st && h_cost && d_i420_fenc && d_i420_ref && slot_fenc >= 0 && slot_fenc < st->slots && (plane == 1 || plane == 2));
    ARG_TRY(!on || (denom >= 0 && denom <= 7 && scale >= -128 && scale <= 127 && offset >= -128 && offset <= 127));
    hipStream_t s = (hipStream_t)stream;
    const size_t S = st->streams;
    StChromaK k;
    memset(&k, 0, sizeof(k));
    k.w = st->w; k.h = st->h; k.pw = st->w / 2; k.ph = st->h / 2; k.bw = st->bw; k.nb = st->nb; k.plane = plane; k.on = on; k.scale = scale; k.denom = denom; k.offset = offset;
    k.i420 = d_i420_fenc; k.ref = d_i420_ref; k.i420_bytes = (size_t)st->w * st->h * 3 / 2; k.out = st->d_acc;
    k.mv = dist > 0 && dist <= st->bframes + 1 && st->searched[slot_fenc][0][dist - 1] ? st->mvs[slot_fenc][0][dist - 1] : nullptr;
    HIP_TRY(hipMemsetAsync(st->d_acc, 0, S * 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_st_weight_cost_chroma, dim3(st->nb, st->streams), dim3(64), 0, s, k);
    std::vector<unsigned long long> v(S * 2);
    HIP_TRY(hipMemcpyAsync(v.data(), st->d_acc, S * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    auto size_ue = [](int x) { int n = 0; x++; while (x >> (n + 1)) n++; return 2 * n + 1; };
    // weight_slice_header_cost, chroma: four times luma's lambda (full resolution), the denominator shared by the two planes
    const int hdr = on ? 4 * st->lambda * (10 + size_ue(denom) + 2 * (size_ue(scale > 0 ? 2 * scale - 1 : -2 * scale) + size_ue(offset > 0 ? 2 * offset - 1 : -2 * offset))) : 0;
    for (size_t i = 0; i < S; i++) h_cost[i] = (int64_t)v[2 * i] + hdr;
    return X264GPU_OK;
}

// ---- macroblock-tree (x264 macroblock_tree's building blocks; the host walks the pictures as x264 does) ----
// x264_adaptive_quant_frame's offsets of the picture in `slot` ([streams][blocks] single floats, device memory; NULL = none)
int x264gpu_slicetype_set_aq(x264gpu_slicetype *st, int slot, const float *d_aq, void *stream)
{
    ARG_TRY(st && slot >= 0 && slot < st->slots);
    st->have_aq[slot] = d_aq != nullptr;
    if (d_aq) HIP_TRY(hipMemcpyAsync(st->aq[slot], d_aq, (size_t)st->streams * st->nb * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return X264GPU_OK;
}
// fenc->i_cost_est_aq of a costed triple (x264 slicetype_mb_cost's i_mb_cost_aq summed over the blocks that count): per stream into h_score[streams]
struct StAqK { int bw, bh, nb, is_i; const int *intra_cost; const uint16_t *lowres_costs; const float *aq; int32_t *out; };
__global__ __launch_bounds__(256) void k_st_cost_aq(StAqK k)
{
    const int s = blockIdx.x;
    int sum = 0;
    for (int i = threadIdx.x; i < k.nb; i += 256) {
        const int bx = i % k.bw, by = i / k.bw;
        const bool score = (bx > 0 && bx < k.bw - 1 && by > 0 && by < k.bh - 1) || k.bw <= 2 || k.bh <= 2;
        if (!score) continue;
        const size_t o = (size_t)s * k.nb + i;
        const int c = k.is_i ? k.intra_cost[o] : k.lowres_costs[o] & LOWRES_COST_MASK;
        sum += (c * st_inv_qscale(k.aq ? k.aq[o] : 0.f) + 128) >> 8;
    }
    __shared__ int red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) red[threadIdx.x] += red[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) k.out[s] = red[0];
}
int x264gpu_slicetype_cost_aq(x264gpu_slicetype *st, int slot, int d0, int d1, int32_t *h_score, void *stream)
{
    ARG_TRY(st && h_score && slot >= 0 && slot < st->slots && d0 >= 0 && d1 >= 0 && d0 <= st->bframes + 1 && d1 <= st->bframes + 1);
    const int nd = st->bframes + 2;
    ARG_TRY(st->cost_est[slot][(size_t)(d0 * nd + d1) * st->streams] >= 0);
    StAqK k;
    memset(&k, 0, sizeof(k));
    k.bw = st->bw; k.bh = st->bh; k.nb = st->nb; k.is_i = d0 == 0 && d1 == 0;
    k.intra_cost = st->intra_cost[slot];
    k.lowres_costs = st->lowres_costs[slot] + (size_t)(d0 * nd + d1) * st->streams * st->nb;
    k.aq = st->have_aq[slot] ? st->aq[slot] : nullptr;
    k.out = st->sums;                          // (the frame-cost calls' scratch sums: every call reads its results back before it returns)
    hipLaunchKernelGGL(k_st_cost_aq, dim3(st->streams), dim3(256), 0, (hipStream_t)stream, k);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_score, st->sums, (size_t)st->streams * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return X264GPU_OK;
}
// --b-bias (x264 i_bframe_bias, -90 .. 100): B costs are scaled by 100 / (120 + bias); set once, before the first cost
int x264gpu_slicetype_set_row_mode(x264gpu_slicetype *st, int serial)
{
    ARG_TRY(st && serial >= -1 && serial <= 1);
    st->serial_rows = serial;
    return X264GPU_OK;
}

int x264gpu_slicetype_set_bframe_bias(x264gpu_slicetype *st, int bias)
{
    ARG_TRY(st && bias >= -90 && bias <= 100);
    st->bframe_bias = bias;
    return X264GPU_OK;
}
int x264gpu_slicetype_clear_propagate(x264gpu_slicetype *st, int slot, void *stream)
{
    ARG_TRY(st && slot >= 0 && slot < st->slots);
    HIP_TRY(hipMemsetAsync(st->prop[slot], 0, (size_t)st->streams * st->nb * sizeof(int32_t), (hipStream_t)stream));
    return X264GPU_OK;
}
// macroblock_tree_propagate(p0, p1, b, referenced): picture b hands its explained cost to p0 (and p1); the costs of the triple must have been computed
int x264gpu_slicetype_propagate(x264gpu_slicetype *st, int s0, int s1, int sb, int d0, int d1, int referenced, void *stream)
{
    ARG_TRY(st && s0 >= 0 && s0 < st->slots && s1 >= 0 && s1 < st->slots && sb >= 0 && sb < st->slots && d0 >= 1 && d1 >= 0 && d0 <= st->bframes + 1 && d1 <= st->bframes + 1);
    const int nd = st->bframes + 2;
    ARG_TRY(st->cost_est[sb][(size_t)(d0 * nd + d1) * st->streams] >= 0);
    StTreeK k;
    memset(&k, 0, sizeof(k));
    k.bw = st->bw; k.bh = st->bh; k.nb = st->nb; k.d0 = d0; k.d1 = d1; k.referenced = referenced;
    const int dsf = d1 > 0 ? ((d0 << 8) + ((d0 + d1) >> 1)) / (d0 + d1) : 256;
    k.bipw0 = st->weightb && d1 > 0 ? 64 - (dsf >> 2) : 32; k.bipw1 = 64 - k.bipw0;
    k.intra_cost = st->intra_cost[sb];
    k.lowres_costs = st->lowres_costs[sb] + (size_t)(d0 * nd + d1) * st->streams * st->nb;
    k.mv[0] = st->mvs[sb][0][d0 - 1]; k.mv[1] = d1 > 0 ? st->mvs[sb][1][d1 - 1] : nullptr;
    k.aq = st->have_aq[sb] ? st->aq[sb] : nullptr;
    k.prop_b = st->prop[sb]; k.prop_ref[0] = st->prop[s0]; k.prop_ref[1] = st->prop[s1];
    hipLaunchKernelGGL(k_st_propagate, dim3((st->nb + 255) / 256, st->streams), dim3(256), 0, (hipStream_t)stream, k);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}
// macroblock_tree_finish: d_out[streams][blocks] = aq - strength * (x264_log2(intra + propagated) - x264_log2(intra) + weightdelta) of the picture in `slot`
int x264gpu_slicetype_finish(x264gpu_slicetype *st, int slot, float strength, float weightdelta, float *d_out, void *stream)
{
    ARG_TRY(st && d_out && slot >= 0 && slot < st->slots && st->intra_calculated[slot]);
    StTreeK k;
    memset(&k, 0, sizeof(k));
    k.bw = st->bw; k.bh = st->bh; k.nb = st->nb; k.strength = strength; k.weightdelta = weightdelta;
    k.intra_cost = st->intra_cost[slot]; k.aq = st->have_aq[slot] ? st->aq[slot] : nullptr; k.prop_b = st->prop[slot]; k.out = d_out;
    hipLaunchKernelGGL(k_st_finish, dim3((st->nb + 255) / 256, st->streams), dim3(256), 0, (hipStream_t)stream, k);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}
const int32_t *x264gpu_slicetype_propagate_cost(x264gpu_slicetype *st, int slot) { return st && slot >= 0 && slot < st->slots ? st->prop[slot] : nullptr; }

int x264gpu_slicetype_intra_mbs(x264gpu_slicetype *st, int slot, int d0, int stream_idx)
{
    if (!st || slot < 0 || slot >= st->slots || d0 < 0 || d0 > st->bframes + 1 || stream_idx < 0 || stream_idx >= st->streams) return -1;
    return st->intra_mbs[slot][(size_t)d0 * st->streams + stream_idx];
}
int x264gpu_slicetype_cost_est(x264gpu_slicetype *st, int slot, int d0, int d1, int stream_idx)
{
    if (!st || slot < 0 || slot >= st->slots || d0 < 0 || d1 < 0 || d0 > st->bframes + 1 || d1 > st->bframes + 1 || stream_idx < 0 || stream_idx >= st->streams) return -1;
    return st->cost_est[slot][(size_t)(d0 * (st->bframes + 2) + d1) * st->streams + stream_idx];
}
// device pointers ([streams][blocks]...) of a picture's cached search results and costs; NULL when that search has not run
const int16_t *x264gpu_slicetype_lowres_mvs(x264gpu_slicetype *st, int slot, int list, int dist)
{
    if (!st || slot < 0 || slot >= st->slots || list < 0 || list > 1 || dist < 1 || dist > st->bframes + 1 || !st->searched[slot][list][dist - 1]) return nullptr;
    return st->mvs[slot][list][dist - 1];
}
const int *x264gpu_slicetype_lowres_mv_costs(x264gpu_slicetype *st, int slot, int list, int dist)
{
    if (!st || slot < 0 || slot >= st->slots || list < 0 || list > 1 || dist < 1 || dist > st->bframes + 1 || !st->searched[slot][list][dist - 1]) return nullptr;
    return st->mvcosts[slot][list][dist - 1];
}
const int *x264gpu_slicetype_intra_costs(x264gpu_slicetype *st, int slot) { return st && slot >= 0 && slot < st->slots && st->intra_calculated[slot] ? st->intra_cost[slot] : nullptr; }
const uint16_t *x264gpu_slicetype_lowres_costs(x264gpu_slicetype *st, int slot, int d0, int d1)
{
    if (!st || slot < 0 || slot >= st->slots || d0 < 0 || d1 < 0 || d0 > st->bframes + 1 || d1 > st->bframes + 1) return nullptr;
    return st->lowres_costs[slot] + (size_t)(d0 * (st->bframes + 2) + d1) * st->streams * st->nb;
}

}  // extern "C"

// trellis.hip.h — x264's CABAC trellis quantiser on the device ([x264-upstream] encoder/rdo.c quant_trellis_cabac; `--trellis 1`: the
// final encode of a macroblock).  Mirrors oracle/trellis.cpp decision for decision; the algorithm is stated in oracle/TRELLIS_NOTES.md.
//
// EIGHT LANES PER BLOCK = THE EIGHT NODES of the search (the abs-level context states CABAC can be in), eight blocks per pass of a
// wavefront.  The coefficients of the blocks sit in LDS in scan order.  Quantised spectra are sparse, so a block does not visit every scan
// position: one step of the loop takes every block from one of ITS non-zero positions (of the round-to-nearest guess) to the next — the
// zeros in between only cost the all-zero node their "significant = 0" bits, one subtraction from a running sum prepared per call.  A step:
//   * every lane, as a SOURCE node, prices its two candidate levels (the round-to-nearest guess q and q - 1) on its own copy of the
//     four context variables a path can touch twice — no traffic between lanes;
//   * every lane, as a DESTINATION node, takes the minimum over the sources that lead to it.  x264's "first strictly better wins" in its
//     evaluation order (level q - 1 before q, sources ascending) is the minimum of (score << 4 | order);
//   * the winner's context bytes and its path (two bits per position: 0 = level 0, 1 = q - 1, 2 = q) come over with one shuffle each.
// The significance / last costs of a position are the same for all blocks (the slice's context variables are only read): lane p works out
// position p's once per call, a block fetches the ones of its current position with a ds_bpermute.  Used by the RD = 3 / 4 instantiations
// of the macroblock loop (k_mb.hip.h: trellis_run) and, as a primitive, by x264gpu_trellis_blocks (tests/test_gpu_prims.py).
#pragma once
#include "cabac_rd.hip.h"

namespace x264gpu {

// x264_cabac_size_unary / x264_cabac_transition_unary ([15][128], context variable = (pStateIdx << 1) | valMPS): built on the host from the
// entropy table and the state transitions (trellis_tables() in prim_kernels.hip), read here through these pointers
struct TrellisTab { const uint16_t *size_unary; const uint8_t *trans_unary; const int *lambda2; };      // lambda2[intra * 52 + qp]: x264_trellis_lambda2_tab
// ... and directly behind lambda2's 104 entries (one allocation, prim_kernels.hip trellis_tables): { mf, bias, unq, w } of every coefficient class at
// every quantiser — row qp holds classes 0..2 of 4x4 blocks, 3..8 of 8x8 blocks, 9 of DC blocks
#define TRELLIS_QT_ROW 40

static __constant__ const uint16_t c_quant4_scale[6][3] = { { 13107, 8066, 5243 }, { 11916, 7490, 4660 }, { 10082, 6554, 4194 },
                                                            { 9362, 5825, 3647 },  { 8192, 5243, 3355 },  { 7282, 4559, 2893 } };
static __constant__ const uint16_t c_quant8_scale[6][6] = { { 13107, 11428, 20972, 12222, 16777, 15481 }, { 11916, 10826, 19174, 11058, 14980, 14290 },
                                                            { 10082, 8943, 15978, 9675, 12710, 11985 },   { 9362, 8228, 14913, 8931, 11984, 11259 },
                                                            { 8192, 7346, 13159, 7740, 10486, 9777 },     { 7282, 6428, 11570, 6830, 9118, 8640 } };
static __constant__ const uint8_t c_zigzag4_fwd[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
static __constant__ const uint8_t c_zigzag8_fwd[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };
static __constant__ const uint16_t c_trellis_w4[3] = { 800, 320, 128 };                       // FIX8(3.125), FIX8(1.25), FIX8(0.5)
static __constant__ const uint16_t c_trellis_w8[6] = { 256, 201, 656, 227, 410, 363 };        // FIX8(1.00000, 0.78487, 2.56132, 0.88637, 1.60040, 1.41850)
static __constant__ const uint8_t c_class8[16] = { 0, 3, 4, 3, 3, 1, 5, 1, 4, 5, 2, 5, 3, 1, 5, 1 };

__device__ __forceinline__ int trellis_class(int cat, int i)       // quantiser / weight class of scan position i
{
    if (cat == 5) { const int pos = c_zigzag8_fwd[i]; return c_class8[((pos >> 3) & 3) * 4 + (pos & 3)]; }
    const int pos = c_zigzag4_fwd[i];
    return (pos & 1) + ((pos >> 2) & 1);
}
__device__ __forceinline__ int shift_round_d(int x, int s) { return s <= 0 ? x << -s : (x + (1 << (s - 1))) >> s; }

// cost (1/256 bit) of bin b on context variable st, and the variable after it
__device__ __forceinline__ int tr_ent(uint32_t model, int st, int b)
{
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((st >> 1) << 2, (int)model);
    return ((st & 1) ^ b) ? (t >> 9) & 0x7ff : t & 0x1ff;
}
__device__ __forceinline__ int tr_next(uint32_t model, int st, int b)
{
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((st >> 1) << 2, (int)model);
    const int s = st >> 1, mps = st & 1;
    if (mps ^ b) return ((int)(t >> 20) << 1) | (s == 0 ? mps ^ 1 : mps);
    return (min(s + 1, 62) << 1) | mps;
}
__device__ __forceinline__ int tr_size_ue_big(unsigned v) { return 2 * (31 - __builtin_clz(v + 1)) + 1; }

// Up to eight blocks of category CAT (0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8) whose coefficients lie in LDS
// in scan order, block b at coefs + b * stride (AC blocks: 16 entries, entry 0 = 0); the levels replace them.  nblk <= 8.  lane & 7 = node,
// lane >> 3 = block.  st_sig / st_last / st_abs: accessors of the slice's context variables of the category (wave-uniform arguments).
// Returns the mask of blocks with a non-zero level (wave-uniform).
// The context variables come in `reg`: the role-indexed register of the category (cabac_rd.hip.h: r for categories 0..4 — this category's byte —,
// r8 for 8x8 blocks).  ONE out-of-line copy per category serves every call site: inlined eight times the search cost the macroblock loop
// twice the register spills.
typedef __attribute__((address_space(3))) int16_t lds_i16;
template <int CAT>
__device__ __noinline__ unsigned trellis_blocks(lds_i16 *coefs, int stride, int nblk, int qp, bool intra, uint32_t model, TrellisTab tt, uint32_t reg)
{
    namespace T = x264gpu_cabac;
    const int lane = (int)threadIdx.x & 63;
    constexpr int st_sh = CAT == 5 || CAT == 2 ? 0 : CAT == 1 ? 8 : CAT == 4 ? 16 : 24;
    constexpr int sig0 = CAT == 3 ? 48 : 0, last0 = CAT == 3 ? 52 : 16, abs0 = CAT == 3 ? 55 : 32;
    auto st_abs = [&](int i) { return (int)((__builtin_amdgcn_readlane(reg, abs0 + i) >> st_sh) & 255); };
    constexpr int NC = CAT == 5 ? 64 : CAT == 3 ? 4 : 16, B_AC = (CAT == 1 || CAT == 4) ? 1 : 0, PW = CAT == 5 ? 4 : 1;
    constexpr bool DC = CAT == 0 || CAT == 3;
    const int n = lane & 7, g = lane >> 3, base = lane & ~7;
    const bool blk_on = g < nblk;
    lds_i16 *mine = coefs + g * stride;
    const int lambda2 = tt.lambda2[(intra ? 52 : 0) + qp];
    // quantiser, rounding offset, inverse and distortion weight of every coefficient class (three for 4x4 blocks, six for 8x8, one for DC
    // blocks), worked out once: the divisions stay out of the loop
    constexpr int NCLS = DC ? 1 : CAT == 5 ? 6 : 3;
    int q_mf[NCLS], q_bias[NCLS], q_unq[NCLS], q_w[NCLS];
    {
        // (from the per-quantiser table: the quantiser is wave-uniform, so these are scalar loads)
        const int4 *row = (const int4 *)(tt.lambda2 + 104 + __builtin_amdgcn_readfirstlane(qp) * TRELLIS_QT_ROW) + (DC ? 9 : CAT == 5 ? 3 : 0);
#pragma unroll
        for (int cl = 0; cl < NCLS; cl++) { const int4 e = row[cl]; q_mf[cl] = e.x; q_bias[cl] = e.y; q_unq[cl] = e.z; q_w[cl] = e.w; }
    }
    auto pick = [&](const int (&t)[NCLS], int cl) {
        int r = t[0];
#pragma unroll
        for (int k2 = 1; k2 < NCLS; k2++) r = cl == k2 ? t[k2] : r;
        return r;
    };
    auto cls_of = [&](int i) { return DC ? 0 : trellis_class(CAT, i); };
    auto guess = [&](int c, int i) { const int cl = cls_of(i); return ((pick(q_bias, cl) + abs(c)) * pick(q_mf, cl)) >> 16; };      // |level| of the round-to-nearest quantiser

    // the positions the guess leaves non-zero, per block (the same mask in its eight lanes)
    unsigned long long nzm = 0;
    for (int p = n; p < NC; p += 8) if (blk_on && p >= B_AC && guess(mine[p], p)) nzm |= 1ull << p;
    for (int m = 1; m < 8; m <<= 1) {
        const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)nzm, m), hi32 = CAT == 5 ? (unsigned)__shfl_xor((int)(unsigned)(nzm >> 32), m) : 0u;
        nzm |= ((unsigned long long)hi32 << 32) | lo32;
    }
    const unsigned long long nzm0 = nzm;
    if (!__ballot(nzm != 0)) {                       // the guess leaves nothing in any block: all levels are zero, no search
        if (CAT == 3) { if (blk_on && n < 4) mine[n] = 0; }
        else for (int p = n; p < NC; p += 8) if (blk_on) mine[p] = 0;
        return 0;
    }
    // What a scan position costs, once per call: lane p = position p holds the bits of its significance / last flags (the slice's context
    // variables are only read, so every block sees the same), the class of its coefficient, and a running sum of the "significant = 0"
    // costs — a block then steps from one of ITS non-zero positions to the next and settles the zeros in between with one subtraction
    uint32_t pos_a = 0, pos_t0 = 0;
    unsigned long long pos_z = 0;
    {
        const int p = lane;
        const bool valid = p >= B_AC && p < NC - 1;
        const int sidx = CAT == 5 ? (int)T::cabac_sig8x8[min(p, 62)] : max(p - B_AC, 0) & 15, lidx = CAT == 5 ? (int)T::cabac_last8x8[min(p, 62)] : max(p - B_AC, 0) & 15;
        const int ss = (int)(((uint32_t)__builtin_amdgcn_ds_bpermute((sig0 + sidx) << 2, (int)reg) >> st_sh) & 255);
        const int sl = (int)(((uint32_t)__builtin_amdgcn_ds_bpermute((last0 + (CAT == 3 ? min(lidx, 3) : lidx)) << 2, (int)reg) >> st_sh) & 255);
        const uint32_t ts = (uint32_t)__builtin_amdgcn_ds_bpermute((ss >> 1) << 2, (int)model), tl = (uint32_t)__builtin_amdgcn_ds_bpermute((sl >> 1) << 2, (int)model);
        const int s0 = (ss & 1) ? (ts >> 9) & 0x7ff : ts & 0x1ff, s1 = (ss & 1) ? ts & 0x1ff : (ts >> 9) & 0x7ff;
        const int l0 = (sl & 1) ? (tl >> 9) & 0x7ff : tl & 0x1ff, l1 = (sl & 1) ? tl & 0x1ff : (tl >> 9) & 0x7ff;
        const int cl = p < NC ? cls_of(min(p, NC - 1)) : 0;
        if (valid) { pos_a = (uint32_t)(s1 + l0) | ((uint32_t)(s1 + l1) << 12); pos_t0 = (uint32_t)(((unsigned long long)s0 * (unsigned long long)lambda2) >> 4); }
        pos_a |= (uint32_t)cl << 24;
        // inclusive prefix sum over the positions: pos_t0 < 2^30, so its two 16-bit halves sum without overflow over 64 lanes — two 32-bit DPP
        // scans (row shifts, then the row broadcasts) instead of six 64-bit shuffle rounds through the LDS crossbar
        pos_z = ((unsigned long long)(unsigned)wave_scan_add((int)(pos_t0 >> 16)) << 16) + (unsigned long long)(unsigned)wave_scan_add((int)(pos_t0 & 0xffffu));
    }

    // level_state: the ten abs-level context variables of the category (wave-uniform), packed four to a word
    uint32_t ls[3] = { 0, 0, 0 };
    for (int i = 0; i < 10; i++) ls[i >> 2] |= (uint32_t)st_abs(CAT == 3 && i > 8 ? 8 : i) << (8 * (i & 3));
    auto level_state = [&](int i) { const uint32_t w = i < 4 ? ls[0] : i < 8 ? ls[1] : ls[2]; return (int)((w >> (8 * (i & 3))) & 255); };
    const uint32_t init4 = (uint32_t)level_state(0) | ((uint32_t)level_state(4) << 8) | ((uint32_t)level_state(8) << 16) | ((uint32_t)level_state(9) << 24);
    constexpr int LG_LAST = CAT == 3 ? 8 : 9;

    // what depends on the node only (this lane's n), worked out before the loop: its level-1 / greater-than-one contexts, where their
    // variables sit in the four bytes a path carries, the slice's values of those a path meets once
    const int l1ctx = n < 4 ? n + 1 : 0, lgctx = n < 4 ? 5 : n == 7 ? LG_LAST : n + 2;
    const int sh_l1 = 8 * (l1ctx >> 2), sh_lg = n >= 6 ? 8 * (lgctx - 6) : 0;
    const int ls_l1 = level_state(l1ctx), ls_lg = level_state(lgctx);
    const unsigned long long SMAX = ~0ull, BIAS = 1ull << 50;
    unsigned long long score = n == 0 ? BIAS : SMAX;
    uint32_t cs = 0, path[PW];
    for (int w = 0; w < PW; w++) path[w] = 0;
    bool ctx_hi = false;

    unsigned long long z_before = 0;                 // running sum of the zero costs below the block's previous non-zero position (exclusive)
    bool started = false;
    for (;;) {
        const bool act = blk_on && nzm != 0, go = act;
        if (!__ballot(act)) break;
        const int i = act ? 63 - __builtin_clzll(nzm) : 0;          // this block's next non-zero position, from the top
        nzm &= ~(1ull << i);
        const int paddr = i << 2;
        const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(paddr, (int)pos_a), t0i = (uint32_t)__builtin_amdgcn_ds_bpermute(paddr, (int)pos_t0);
        const unsigned long long zi = ((unsigned long long)(uint32_t)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)(pos_z >> 32)) << 32) | (uint32_t)__builtin_amdgcn_ds_bpermute(paddr, (int)(unsigned)pos_z);
        // the zeros between the previous non-zero position and this one: nothing to choose; the all-zero path of a block still in ctx_lo is
        // spared their significance bits (subtracting from one node is adding to the rest)
        if (act && started && !ctx_hi && n == 0) score -= z_before - zi;
        if (act) { z_before = zi - t0i; started = true; }
        const int cost1 = (int)(pa & 4095), cost2 = (int)((pa >> 12) & 4095), cl = (int)(pa >> 24);
        const unsigned long long cost0_l = t0i;          // (bits of "significant = 0" x lambda2) >> 4 of this position
        const int un = pick(q_unq, cl), wgt = pick(q_w, cl);
        const int c = act ? (int)mine[i] : 0, a = abs(c), q = act ? ((pick(q_bias, cl) + a) * pick(q_mf, cl)) >> 16 : 1;
        // ---- every lane as a source node: its two candidate levels A = q - 1, B = q ----
        unsigned long long ssd0[2], ssd1[2];
        for (int kk = 0; kk < 2; kk++) {
            const int lvl = q - 1 + kk, ua = (int)(((long long)un * lvl + 128) >> 8);
            int d = a - ua;                                                   // |d| < 2^15.5: the square fits 32 bits
            ssd1[kk] = (unsigned long long)(unsigned)(d * d) * (unsigned)wgt;
            ssd0[kk] = ssd1[kk];
            if (i == 0 && !DC && !ctx_hi) { d = c - (((c < 0 ? -ua : ua) + 8) & ~15); ssd0[kk] = (unsigned long long)(unsigned)(d * d) * (unsigned)wgt; }
        }
        const bool alive = (long long)score >= 0;
        const bool src_ok = go && (ctx_hi ? (n >= 1 && alive) : (n <= 3 && (n == 0 || alive)));
        // kinds of the two candidates: 0 = level 0 (a copy), 1 = level 1, 2 = level >= 2
        const int kindA = q == 1 ? 0 : q == 2 ? 1 : 2, kindB = q == 1 ? 1 : 2;
        if (q == 1) ssd1[0] += cost0_l;
        unsigned long long candv[2];
        uint32_t candcs[2];
        // (the model lookups are ds_bpermutes: every lane must take part, whatever its block is doing — so nothing below is skipped, the
        //  kinds only select)
        const int l1state = n >= 3 ? (int)((cs >> sh_l1) & 255) : ls_l1;
        const int lgstate = n >= 6 ? (int)((cs >> sh_lg) & 255) : ls_lg;
        // one lookup of the model gives both bin costs and both successors of the level-1 context variable
        const uint32_t tm = (uint32_t)__builtin_amdgcn_ds_bpermute((l1state >> 1) << 2, (int)model);
        const int l1s = l1state >> 1, l1m = l1state & 1, c_mps = (int)(tm & 0x1ff), c_lps = (int)((tm >> 9) & 0x7ff);
        const int s_mps = (min(l1s + 1, 62) << 1) | l1m, s_lps = ((int)(tm >> 20) << 1) | (l1s == 0 ? l1m ^ 1 : l1m);
        const int ent_l1_0 = l1m ? c_lps : c_mps, ent_l1_1 = l1m ? c_mps : c_lps;
        const int nxt_l1_0 = l1m ? s_lps : s_mps, nxt_l1_1 = l1m ? s_mps : s_lps;
        // ... and the same for the greater-than-one context variable: cost of a zero, of a one, and of a zero after a one — only on steps where
        // some block tries a level of two or more (a wave-uniform branch: every lane takes part in the lookups or none does)
        int lg_c0 = 0, lg_c1 = 0, lg_c10 = 0, lg_n0 = 0, lg_n10 = 0;       // costs, and the variable after "0" / after "1 0"
        const bool any_big = __ballot(go && q >= 2) != 0;
        if (any_big) {
            const uint32_t tg = (uint32_t)__builtin_amdgcn_ds_bpermute((lgstate >> 1) << 2, (int)model);
            const int lgs = lgstate >> 1, lgm = lgstate & 1;
            const int cm = (int)(tg & 0x1ff), cl = (int)((tg >> 9) & 0x7ff), sm = (min(lgs + 1, 62) << 1) | lgm, sl = ((int)(tg >> 20) << 1) | (lgs == 0 ? lgm ^ 1 : lgm);
            lg_c0 = lgm ? cl : cm; lg_c1 = lgm ? cm : cl; lg_n0 = lgm ? sl : sm;
            const int after1 = lgm ? sm : sl;
            const uint32_t t2 = (uint32_t)__builtin_amdgcn_ds_bpermute((after1 >> 1) << 2, (int)model);
            const int a_s = after1 >> 1, a_m = after1 & 1;
            lg_c10 = a_m ? (int)((t2 >> 9) & 0x7ff) : (int)(t2 & 0x1ff);
            lg_n10 = a_m ? ((int)(t2 >> 20) << 1) | (a_s == 0 ? a_m ^ 1 : a_m) : (min(a_s + 1, 62) << 1) | a_m;
        }
        // levels of four and more (prefix >= 3) read x264_rdo_init's tables; rare, so behind a wave-uniform test
        const bool any_huge = __ballot(go && q >= 4) != 0;
        for (int kk = 0; kk < 2; kk++) {
            const int kind = kk ? kindB : kindA, lvl = q - 1 + kk;
            const unsigned long long rel0 = q == 1 ? ssd0[kk] - ssd1[0] : ssd0[kk], rel1 = q == 1 ? ssd1[kk] - ssd1[0] : ssd1[kk];
            const int node_ctx = kind == 1 ? (n < 3 ? n + 1 : n == 3 ? 3 : n) : (n < 4 ? 4 : min(n + 1, 7));
            const int prefix = max(min(lvl - 1, 14), 0);
            unsigned unary = prefix == 1 ? (unsigned)(lg_c0 + 256) : (unsigned)(lg_c1 + lg_c10 + 256);
            int lg_next = prefix == 1 ? lg_n0 : lg_n10;
            if (any_huge && prefix >= 3) { unary = tt.size_unary[prefix * 128 + lgstate]; lg_next = tt.trans_unary[prefix * 128 + lgstate]; }
            const unsigned f8 = (unsigned)(n ? cost1 : cost2) + (unsigned)(kind == 2 ? ent_l1_1 : ent_l1_0)
                                + (kind == 2 ? unary + (lvl >= 15 ? (unsigned)tr_size_ue_big((unsigned)(lvl - 15)) << 8 : 0u) : 256u);
            // trellis_coef0 (level 0): node j -> node j, only node 0 pays a distortion difference (the others carry it in the baseline)
            const unsigned long long v0 = score + (n == 0 ? ssd0[0] - ssd1[0] : 0ull);
            const unsigned long long v12 = score + (n ? rel1 : rel0) + (((unsigned long long)f8 * (unsigned long long)lambda2) >> 4);
            const unsigned long long v = kind == 0 ? v0 : v12;
            uint32_t ncs = (n == 2 || (n <= 3 && node_ctx == 4)) ? init4 : cs;
            if (n >= 3) ncs = (ncs & ~(255u << sh_l1)) | ((uint32_t)(kind == 2 ? nxt_l1_1 : nxt_l1_0) << sh_l1);
            if (kind == 2 && n >= 6) ncs = (ncs & ~(255u << sh_lg)) | ((uint32_t)lg_next << sh_lg);          // (node 7 is where nodes 6 and 7 go)
            // the key a destination compares: score, then x264's evaluation order (level q - 1 before q, sources ascending) as the tie-break
            candv[kk] = src_ok ? (v << 4) | (unsigned)(kk * 8 + n) : ~0ull; candcs[kk] = kind == 0 ? cs : ncs;
        }
        // ---- every lane as a destination node n: the sources that lead here, in x264's evaluation order.  They sit at n, n - 1, .. n - 4 of the
        //      own block: DPP row shifts bring their keys over without a trip through LDS (a shift that crosses into the block before is never
        //      a valid source) ----
        unsigned long long best = ~0ull;
        auto shr = [&](unsigned long long key, auto tag) {
            constexpr int CTRL = 0x110 + decltype(tag)::value;            // row_shr:N
            const unsigned lo32 = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)key, CTRL, 0xf, 0xf, false);
            const unsigned hi32 = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(key >> 32), CTRL, 0xf, 0xf, false);
            return ((unsigned long long)hi32 << 32) | lo32;
        };
        for (int kk = 0; kk < 2; kk++) {
            const int kind = kk ? kindB : kindA;
            const unsigned long long k0 = candv[kk], k1 = shr(candv[kk], std::integral_constant<int, 1>{});
            // level 0: n <- n.  level 1: 1 <- 0, 2 <- 1, 3 <- 2 and 3, n >= 4 <- n.  levels >= 2: 4 <- 0..3, 5 <- 4, 6 <- 5, 7 <- 6 and 7
            const bool self_ok = kind == 0 || (kind == 1 ? n >= 3 : n == 7);
            const bool m1_ok = kind == 1 ? (n >= 1 && n <= 3) : kind == 2 && n >= 4;
            if (m1_ok && k1 < best) best = k1;            // (the lower source first: its order value is lower anyway)
            if (self_ok && k0 < best) best = k0;
            if (any_big) {
                const unsigned long long k2 = shr(candv[kk], std::integral_constant<int, 2>{}), k3 = shr(candv[kk], std::integral_constant<int, 3>{}), k4 = shr(candv[kk], std::integral_constant<int, 4>{});
                if (kind == 2 && n == 4) { if (k2 < best) best = k2; if (k3 < best) best = k3; if (k4 < best) best = k4; }
            }
        }
        const bool won = best != ~0ull;
        const int wj = (int)(best & 7), wk = (int)((best >> 3) & 1);
        const int waddr = (base + wj) << 2;
        const uint32_t csA = (uint32_t)__builtin_amdgcn_ds_bpermute(waddr, (int)candcs[0]), csB = (uint32_t)__builtin_amdgcn_ds_bpermute(waddr, (int)candcs[1]);
        uint32_t npath[PW];
        for (int w = 0; w < PW; w++) npath[w] = (uint32_t)__builtin_amdgcn_ds_bpermute(waddr, (int)path[w]);
        if (go) {
            score = won ? best >> 4 : SMAX;
            cs = wk ? csB : csA;
            const uint32_t choice = won ? (uint32_t)(wk + 1) : 0u;       // 1 = level q - 1, 2 = level q  (q - 1 == 0 codes as level 0 by value)
            for (int w = 0; w < PW; w++) { path[w] = npath[w]; if (w == (i >> 4)) path[w] |= choice << (2 * (i & 15)); }
            if (q >= 2) ctx_hi = true;
        }
    }
    if (blk_on && started && !ctx_hi && n == 0) score -= z_before;          // the zeros below the lowest non-zero position
    // ---- the best node of every block; node 0 = nothing left ----
    const bool cand = ctx_hi ? n >= 1 : n <= 3;
    unsigned long long key = cand && (long long)score >= 0 ? (score << 3) | (unsigned)n : ~0ull;
    for (int m = 1; m < 8; m <<= 1) {
        const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)key, m), hi32 = (unsigned)__shfl_xor((int)(unsigned)(key >> 32), m);
        const unsigned long long o = ((unsigned long long)hi32 << 32) | lo32;
        key = o < key ? o : key;
    }
    const int bn = (int)(key & 7);
    uint32_t bpath[PW];
    for (int w = 0; w < PW; w++) bpath[w] = (uint32_t)__shfl((int)path[w], base + bn);
    // ---- levels out: lane n writes positions n, n + 8, ... ----
    int lv[NC / 8 > 0 ? NC / 8 : 1];
    bool nz = false;
    for (int t = 0, p = n; p < NC; p += 8, t++) {
        const int c = blk_on ? (int)mine[p] : 0, q = guess(c, p);
        const uint32_t choice = (bpath[CAT == 5 ? p >> 4 : 0] >> (2 * (p & 15))) & 3;
        int l = 0;
        if (blk_on && bn != 0 && ((nzm0 >> p) & 1) && choice) l = choice == 1 ? q - 1 : q;
        lv[t] = c < 0 ? -l : l;
        nz = nz || lv[t] != 0;
    }
    __builtin_amdgcn_wave_barrier();
    if (CAT == 3) { if (blk_on && n < 4) mine[n] = (int16_t)lv[0]; }
    else for (int t = 0, p = n; p < NC; p += 8, t++) if (blk_on) mine[p] = (int16_t)lv[t];
    const unsigned long long nzb = __ballot(nz);
    unsigned out = 0;
    for (int b = 0; b < 8; b++) if ((nzb >> (8 * b)) & 0xff) out |= 1u << b;
    return out;
}

}  // namespace x264gpu

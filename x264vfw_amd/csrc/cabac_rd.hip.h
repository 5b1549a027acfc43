// cabac_rd.hip.h — CABAC as x264's rate-distortion code needs it on the device (k_mb.hip.h, RD instantiations of CABAC sessions).
//
// x264 ([x264-upstream] encoder/cabac.c + encoder/rdo.c behind x264_encoder_encode, reference call site codec.c:1693) keeps one set of
// context variables per slice, moved on by the entropy coding of every finished macroblock, and prices a candidate by running the
// macroblock's syntax over a COPY of them with the arithmetic coder replaced by a counter in 1/256 bit (entropy[state ^ bin], bypass
// bin = 256, the I16x16 terminate bin = 7).  The bitstream itself is still written by the host (host/cabac.cpp); what lives here is
//   * the context variables of the slice a wavefront codes, ONE BYTE EACH IN THREE VGPRs (cab_locate below) — copying them for a candidate
//     is three moves.  The header syntax and the coded_block_flags are coded by wave-uniform code that reads and writes its register with
//     v_readlane / a lane select; the residual contexts are laid out so that EVERY LANE OWNS ONE CONTEXT of the block category being
//     coded: the significance map of a 4x4 block is one step of all lanes, and the serial part of a block is one step per non-zero
//     coefficient (per-context bin order is all the arithmetic model cares about, and bits add);
//   * a register with the probability model: lane s = cost of the MPS | cost of the LPS << 9 | state after an LPS << 20;
//   * cab_mb(): the macroblock layer's bins, either in bitstream order ("evolve": what the finished macroblock leaves behind, skip flag
//     included) or as x264_macroblock_size_cabac walks them ("size": no skip flag, residual blocks from the last coefficient down with
//     flags and levels interleaved — the order matters for 8x8 blocks, whose positions share contexts).
// Mirrors oracle/cabac_rd.cpp bin for bin; the initialisation values and state transitions are the host coder's tables.
#pragma once
#include <type_traits>
#include "enc_common.hip.h"
#include "cabac_layout.hip.h"
#define CABAC_TABLE static __constant__ const
#define CABAC_NAMESPACE x264gpu_cabac
#include "../host/cabac_tables.hpp"
#undef CABAC_TABLE
#undef CABAC_NAMESPACE

#ifndef CAB_WALK_BINS
#define CAB_WALK_BINS 8          /* bins a chain-table lookup takes (the table holds 0..8): fewer = a smaller hot part of the table */
#endif
namespace x264gpu {

static __constant__ const uint16_t c_cabac_entropy[128] = {
#include "cabac_entropy.inc"
};

struct Cab { uint32_t a, r, r8; int f8; int f8v; };       // f8: wave-uniform count, f8v: this lane's share of the parallel steps (1/256 bit)

__device__ __forceinline__ uint32_t cab_model(int lane)
{
    return (uint32_t)c_cabac_entropy[2 * lane] | ((uint32_t)c_cabac_entropy[2 * lane + 1] << 9) | ((uint32_t)x264gpu_cabac::cabac_trans_lps[lane] << 20);
}

__device__ __forceinline__ int cab_total(const Cab &cb) { return cb.f8 + wave_sum(cb.f8v); }

// 9.3.1.1: context variables from the slice quantiser (cabac_init_idc 0)
__device__ __forceinline__ void cab_init(Cab &cb, int lane, bool pslice, int qp)
{
    namespace T = x264gpu_cabac;
    qp = min(max(qp, 0), 51);
    uint32_t w[3] = { 0, 0, 0 };
    auto seed = [&](int ctx) {
        int m = 0, n = 0;
        if (ctx < 276) { const T::CabacInitRow row = T::cabac_init_0_275[ctx]; m = pslice ? row.mp : row.mi; n = pslice ? row.np : row.ni; }
        else { const T::CabacInitRow row = T::cabac_init_399_435[ctx - 399]; m = pslice ? row.mp : row.mi; n = pslice ? row.np : row.ni; }
        const int pre = min(max(((m * qp) >> 4) + n, 1), 126);
        return (uint32_t)(pre <= 63 ? (63 - pre) << 1 : ((pre - 64) << 1) | 1);
    };
    for (int j = 0; j < 4; j++) {
        const int c = lane * 4 + j;
        if (c < 105) w[0] |= seed(c) << (8 * j); else if (c < 108) w[0] |= seed(399 + c - 105) << (8 * j);
    }
    {
        const int role = lane >> 4, i = lane & 15;
        const int sig_off[5] = { 105, 120, 134, 149, 152 }, last_off[5] = { 166, 181, 195, 210, 213 }, abs_off[5] = { 227, 237, 247, 257, 266 }, n1[5] = { 15, 14, 15, 3, 14 };
        const int cats[3] = { 2, 1, 4 };
        for (int b = 0; b < 3; b++) {
            const int cat = cats[b];
            if (role == 0 && i < n1[cat]) w[1] |= seed(sig_off[cat] + i) << (8 * b);
            else if (role == 1 && i < n1[cat]) w[1] |= seed(last_off[cat] + i) << (8 * b);
            else if (role == 2 && i < 10) w[1] |= seed(abs_off[cat] + i) << (8 * b);
        }
        if (lane < 48) {
            if (role == 0 && i < 15) w[1] |= seed(105 + i) << 24; else if (role == 1 && i < 15) w[1] |= seed(166 + i) << 24; else if (role == 2 && i < 10) w[1] |= seed(227 + i) << 24;
        } else {
            const int l = lane - 48;
            if (l < 3) w[1] |= seed(149 + l) << 24; else if (l >= 4 && l < 7) w[1] |= seed(210 + l - 4) << 24; else if (l >= 7) w[1] |= seed(257 + l - 7) << 24;
        }
        if (role == 0 && i < 15) w[2] = seed(402 + i); else if (role == 1 && i < 9) w[2] = seed(417 + i); else if (role == 2 && i < 10) w[2] = seed(426 + i);
    }
    cb.a = w[0]; cb.r = w[1]; cb.r8 = w[2]; cb.f8 = 0; cb.f8v = 0;
}

// one bin of the header syntax / a coded_block_flag (contexts 0..104, 399..401): wave-uniform, context variables in register a
__device__ __forceinline__ void cab_bin(Cab &cb, uint32_t model, int lane, int ctx, int bin)
{
    ctx = __builtin_amdgcn_readfirstlane(ctx >= 399 ? ctx - 399 + 105 : ctx); bin = __builtin_amdgcn_readfirstlane(bin);
    const int li = ctx >> 2, sh = (ctx & 3) * 8;
    const uint32_t w = __builtin_amdgcn_readlane(cb.a, li);
    const int st = (w >> sh) & 255, sg = st >> 1, mps = st & 1;
    const uint32_t t = __builtin_amdgcn_readlane(model, sg);
    const bool lps = (mps ^ bin) != 0;
    cb.f8 += lps ? (t >> 9) & 0x7ff : t & 0x1ff;
    const int ns = lps ? (int)(t >> 20) : min(sg + 1, 62);
    const int nm = lps && sg == 0 ? mps ^ 1 : mps;
    const uint32_t nw = (w & ~(255u << sh)) | ((uint32_t)((ns << 1) | nm) << sh);
    cb.a = lane == li ? nw : cb.a;
}
__device__ __forceinline__ void cab_bypass(Cab &cb, int n = 1) { cb.f8 += 256 * n; }
__device__ __forceinline__ void cab_ue_bypass(Cab &cb, int k, int v)
{
    int n = 0;
    while (v >= (1 << k)) { n++; v -= 1 << k; k++; }
    cb.f8 += 256 * (n + 1 + k);
}

// one bin in every lane that has one (mine): the lane's own context variable st, its share of the bits
__device__ __forceinline__ void cab_step(int &st, int &f8v, uint32_t model, bool mine, int bin)
{
    const int sg = st >> 1, mps = st & 1;
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute(sg << 2, (int)model);
    const bool lps = (mps ^ bin) != 0;
    const int cost = lps ? (t >> 9) & 0x7ff : t & 0x1ff;
    const int ns = lps ? (int)(t >> 20) : min(sg + 1, 62);
    const int nm = lps && sg == 0 ? mps ^ 1 : mps;
    f8v += mine ? cost : 0;
    st = mine ? (ns << 1) | nm : st;
}

// ---- coeff_abs_level_minus1 of EVERY block of a macroblock (or part) in one walk ----
// The bins a coefficient sends depend, besides the context variables, only on its own block: x264's node (how many ones / greater-than-ones were
// coded before it, i.e. at the higher scan positions) picks the context c1 of its first bin and the context cg of the bins after it.  The arithmetic
// model only cares about the order of bins WITHIN a context and bits add, so the walk goes CONTEXT BY CONTEXT instead of coefficient by coefficient,
// and a context's chain is wave-uniform scalar code over ballot masks — no trip through the LDS crossbar per bin:
//   * per word of 64 coefficient slots (four 4x4 blocks or one 8x8 block of luma, the blocks of one chroma plane, a DC block; slot order = coding
//     order: blocks ascending, scan positions descending) every lane looks at its slot's coefficient and works out its node with two ballots;
//   * a c1 context's bins are the "level > 1" flags of ITS coefficients (a ballot mask): the bins equal to the context's more probable symbol form runs
//     that cost a difference of prefix sums over the states (lane s of the model register IS state s) and move the state up by their length — one
//     step per LESS probable symbol, not per bin;
//   * a cg context's bins are the unary strings of its coefficients above one: a run of ones on a context whose more probable symbol is one is the
//     same closed form.
// Measured before this form (tools/mb_prof.py, -DMB_PROF_RD): one step per coefficient through ds_bpermute cost 400 - 660 cycles a coefficient.
struct CabLv {
    int cat0;                // luma category of the blocks below: 2 (4x4), 5 (8x8), 1 (the AC blocks of an Intra_16x16), -1 none
    unsigned nz0;            // luma blocks with coefficients (bit = block; 8x8: bit = 8x8 block)
    unsigned nzac;           // chroma AC blocks with coefficients (bit = plane * 4 + block)
    unsigned nzdc;           // chroma DC blocks with coefficients (bit = plane)
    bool ldc;                // the luma DC block of an Intra_16x16 has coefficients
};

// inclusive prefix sum of v over the wavefront's 64 lanes (DPP: row_shr 1 / 2 / 4 / 8 inside each row of sixteen, then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3; lanes without a source add the identity)
__device__ __forceinline__ int wave_scan_add(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
__device__ __forceinline__ int cab_mbcnt(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// The chain table (prim_kernels.hip cabac_chain_table): entry ((2^k - 1) + pattern) * 128 + variable = what k = 0..8 bins (first bin = bit 0 of the
// pattern) leave of a context variable (bits 0..6) and what they cost (1/256 bit, above); k = 0: the variable itself, no cost
#define CAB_CHAIN_ENTRIES (511 * 128)
// v_writelane_b32: a wave-uniform value into ONE lane of a register.  This clang has no builtin for it; the lane is an inline constant (two scalar
// registers in one VALU instruction would break the constant-bus limit).  gfx940+ needs two wait states between a VALU write of an SGPR / VCC (the
// v_cmp behind a ballot) and a VALU read of it, and the compiler's hazard recogniser does not look inside inline assembly — measured on MI355X:
// without the s_nop the lane receives what the register held BEFORE the compare (HISTORY.md §9)
template <int LN>
__device__ __forceinline__ void cab_writelane3(unsigned &lo, unsigned &hi, int &n, unsigned long long val, int nval)
{
    asm("s_nop 1\n\tv_writelane_b32 %0, %3, %6\n\tv_writelane_b32 %1, %4, %6\n\tv_writelane_b32 %2, %5, %6"
        : "+v"(lo), "+v"(hi), "+v"(n) : "s"((unsigned)val), "s"((unsigned)(val >> 32)), "s"(nval), "n"(LN));
}
// one context's chain, wave-uniform: state index sg, more probable symbol mps, bits into f8
struct CabChain { int sg, mps; };
// n bins equal to the more probable symbol: every bin moves the state up by one (to 62 at most) and costs what the model says for the state it leaves
__device__ __forceinline__ void cab_run_mps(CabChain &c, int &f8, int cumv, int m62, int n)
{
    const int hi = min(c.sg + n, 62);
    f8 += __builtin_amdgcn_readlane(cumv, hi) - __builtin_amdgcn_readlane(cumv, c.sg) + (c.sg + n - hi) * m62;
    c.sg = hi;
}
// one bin of the less probable symbol
__device__ __forceinline__ void cab_one_lps(CabChain &c, int &f8, uint32_t model)
{
    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)model, c.sg);
    f8 += (int)((t >> 9) & 0x7ff);
    c.mps ^= c.sg == 0 ? 1 : 0;
    c.sg = (int)(t >> 20);
}

// the "level > 1" flags of the coefficients whose node picks context 0 .. 4, each context's compacted into a bit string: every member pushes its flag
// to the lane of its rank among the members (ds_permute; non-members push a zero to lane 63 — a member of rank 63 exists only when every lane is
// one), a ballot reads the string off, and it goes into lane LANE0 + V.  The five pushes go out together (one trip's latency, not five)
template <int LANE0, class PROF>
__device__ __forceinline__ void cab_strings_c1(int a, int c1, unsigned &blo, unsigned &bhi, int &N, PROF &pf)
{
    unsigned long long cur[5];
    int pushed[5];
#pragma unroll
    for (int v = 0; v < 5; v++) {
        const bool member = a != 0 && c1 == v;
        cur[v] = __ballot(member);
        pushed[v] = __builtin_amdgcn_ds_permute((member ? cab_mbcnt(cur[v]) : 63) << 2, member && a > 1 ? 1 : 0);
    }
    if (cur[0]) { pf.count(20); cab_writelane3<LANE0 + 0>(blo, bhi, N, __ballot(pushed[0] != 0), __builtin_popcountll(cur[0])); }
    if (cur[1]) { pf.count(20); cab_writelane3<LANE0 + 1>(blo, bhi, N, __ballot(pushed[1] != 0), __builtin_popcountll(cur[1])); }
    if (cur[2]) { pf.count(20); cab_writelane3<LANE0 + 2>(blo, bhi, N, __ballot(pushed[2] != 0), __builtin_popcountll(cur[2])); }
    if (cur[3]) { pf.count(20); cab_writelane3<LANE0 + 3>(blo, bhi, N, __ballot(pushed[3] != 0), __builtin_popcountll(cur[3])); }
    if (cur[4]) { pf.count(20); cab_writelane3<LANE0 + 4>(blo, bhi, N, __ballot(pushed[4] != 0), __builtin_popcountll(cur[4])); }
}
// the unary strings of the coefficients above one on context V (.. 9), laid end to end: all ones but for the zero that ends the string of a level below
// fifteen.  Where a string starts is a prefix sum of the lengths; every member with a closing zero pushes a flag to the lane of that zero's position
// and a ballot gives the zeros — when the word's strings fit 63 bins (else: bit V - 5 of `serial`)
template <int LANE0, int V, class PROF>
__device__ __forceinline__ void cab_strings_cg(int a, int a15, int cg, unsigned &blo, unsigned &bhi, int &N, unsigned long long &serial, PROF &pf)
{
    const bool member = a > 1 && cg == V;
    if (__ballot(member)) {
        pf.count(20);
        const int incl = wave_scan_add(member ? (a15 < 15 ? a15 - 1 : 13) : 0);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total <= 63) {
            const bool closes = member && a15 < 15;
            const int pushed = __builtin_amdgcn_ds_permute((closes ? incl - 1 : 63) << 2, closes ? 1 : 0);      // (the zero is the string's last bin)
            cab_writelane3<LANE0 + V>(blo, bhi, N, ((1ull << total) - 1ull) & ~__ballot(pushed != 0), total);
        } else serial |= 1ull << (V - 5);
    }
    if constexpr (V < 9 && LANE0 + V + 1 < 64) cab_strings_cg<LANE0, V + 1>(a, a15, cg, blo, bhi, N, serial, pf);
}

template <class PROF>
__device__ __forceinline__ void cab_levels_all(Cab &cb, uint32_t model, int lane, const int16_t *lvs, const CabLv &W, const uint32_t *ctab, PROF &pf)
{
    if (W.cat0 < 0 && !W.nzac && !W.nzdc && !W.ldc) return;
    lane = relane(lane);
    const int cumv = wave_scan_add((int)(model & 0x1ff)) - (int)(model & 0x1ff);          // lane s: the more probable symbol's cost summed over the states below s
    const int m62 = (int)(__builtin_amdgcn_readlane((int)model, 62) & 0x1ff);
    int f8 = 0;
    // one group of blocks that share their ten contexts: its words of 64 slots one after the other.  a_of(w): this lane's |level| in word w (0: no
    // coefficient); gshift: log2 of the slots of a block; the contexts sit in lanes lane0 + q of register reg, byte sh
    auto group = [&](auto a_of, int nwords, unsigned wordmask, int gshift, uint32_t &reg, int sh, auto lane0_tag, int bac, int ncm1, int sigbase, int lastbase) {
        constexpr int LANE0 = decltype(lane0_tag)::value;
        for (int w = 0; w < nwords; w++) {
            if (!((wordmask >> w) & 1u)) continue;
            const int a = a_of(w);
            const unsigned long long nzm = __ballot(a != 0);
            if (!nzm) continue;
            const unsigned long long m1 = __ballot(a == 1), mg = nzm & ~m1;
            int ngt, n1;                     // coded before this coefficient: the lower slots of its block
            if (gshift == 6) { ngt = cab_mbcnt(mg); n1 = cab_mbcnt(m1); }
            else {
                const int base = lane & ~((1 << gshift) - 1);
                const unsigned sel = (1u << (lane - base)) - 1u;
                ngt = __builtin_popcount((unsigned)(mg >> base) & sel); n1 = __builtin_popcount((unsigned)(m1 >> base) & sel);
            }
            const int node = ngt == 0 ? min(n1, 3) : min(3 + ngt, 7);
            const int c1 = node < 4 ? node + 1 : 0, cg = node < 4 ? 5 : node + 2;
            // sign: one bypass bin; escape: Exp-Golomb order 0 of a - 15 = 2 floor(log2(a - 14)) + 1 bypass bins
            cb.f8v += a ? 256 + (a >= 15 ? 256 * (2 * (31 - __builtin_clz(a - 14)) + 1) : 0) : 0;
            const int a15 = min(a, 15);
            pf.count(21, __builtin_popcountll(nzm));
            // ---- the ten contexts' bin strings, each into the lane that holds the context (lane0 + q): B = the bins (first = bit 0), N = how many ----
            unsigned blo = 0, bhi = 0; int N = 0;
            cab_strings_c1<LANE0>(a, c1, blo, bhi, N, pf);
            unsigned long long serial = 0;                   // the cg contexts (bit q - 5) whose strings do not fit a word: coded coefficient by coefficient below
            if (mg) cab_strings_cg<LANE0, 5>(a, a15, cg, blo, bhi, N, serial, pf);
            // ---- the significance map (4x4-type blocks; an 8x8 block's positions share contexts: cab_block8): position p of a block has its own
            //      significant_coeff_flag context (lane sigbase + p) and last_significant_coeff_flag context (lane lastbase + p), so a context sees one
            //      bin a block at most — every such lane appends ITS bin of each block of the word, straight from the word's non-zero mask ----
            if (ncm1) {
                const int nslots = 1 << gshift;
                const bool is_s = lane >= sigbase && lane < sigbase + ncm1, is_l = lane >= lastbase && lane < lastbase + ncm1;
                const int pp = is_s ? lane - sigbase : lane - lastbase;                      // the position (AC blocks: of the coefficients after the DC)
                const int sbit = min(max(nslots - 1 - (pp + bac), 0), 31);                  // my coefficient's slot inside a block's field of the mask
                for (int j = 0; j < (gshift == 4 ? 4 : gshift == 2 ? 2 : 1); j++) {
                    const unsigned field = (unsigned)(nzm >> (j * nslots)) & ((1u << nslots) - 1u);
                    if (!field) continue;
                    const int lastp = nslots - 1 - __builtin_ctz(field) - bac;              // the block's last non-zero position
                    const unsigned nzb = (field >> sbit) & 1u;
                    const bool cond = is_s ? pp <= lastp : is_l && nzb && pp <= lastp;
                    const unsigned bit = is_s ? nzb : (pp == lastp ? 1u : 0u);
                    blo |= cond ? bit << N : 0u; N += cond ? 1 : 0;
                }
            }
            // ---- every context lane walks ITS string through the chain table, eight bins a lookup; the lanes' lookups go out together ----
            {
                int st = (int)((reg >> sh) & 255u);
                unsigned long long B = ((unsigned long long)bhi << 32) | blo;
                while (__ballot(N > 0)) {
                    const int k = min(N, CAB_WALK_BINS);
                    const uint32_t e = ctab[((((1 << k) - 1) + ((int)(unsigned)B & ((1 << k) - 1))) << 7) + st];
                    st = (int)(e & 127u); cb.f8v += (int)(e >> 7);
                    B >>= k; N -= k;
                }
                reg = (reg & ~(255u << sh)) | ((uint32_t)st << sh);
            }
            // ---- (rare) a cg context with more than 63 bins in one word: its coefficients one by one ----
            for (int v = 5; serial && v < 10; v++) {
                if (!((serial >> (v - 5)) & 1ull)) continue;
                unsigned long long cur = __ballot(a > 1 && cg == v);
                const int stv = (__builtin_amdgcn_readlane((int)reg, LANE0 + v) >> sh) & 255;
                CabChain c = { stv >> 1, stv & 1 };
                while (cur) {
                    pf.count(20);
                    const int f = __builtin_ctzll(cur);
                    cur &= cur - 1ull;
                    const int av = __builtin_amdgcn_readlane(a15, f);
                    int ones = av - 2;                                   // (15 - 2 = the 13 ones of the escape)
                    while (ones > 0) {
                        if (c.mps) { cab_run_mps(c, f8, cumv, m62, ones); ones = 0; }
                        else { cab_one_lps(c, f8, model); ones--; }
                    }
                    if (av != 15) { if (c.mps) cab_one_lps(c, f8, model); else cab_run_mps(c, f8, cumv, m62, 1); }
                }
                const uint32_t nst = (uint32_t)((c.sg << 1) | c.mps);
                reg = lane == LANE0 + v ? (reg & ~(255u << sh)) | (nst << sh) : reg;
            }
        }
    };
    const int g = lane >> 4, q = lane & 15;
    if (W.cat0 == 5)
        group([&](int w) { const int pos = 63 - lane; return abs((int)lvs[(w * 4 + (pos & 3)) * 16 + (pos >> 2)]); }, 4, W.nz0 & 15u, 6, cb.r8, 0, std::integral_constant<int, 32>{}, 0, 0, 0, 0);
    else if (W.cat0 >= 0) {
        const bool ac = W.cat0 == 1;
        const unsigned wm = ((W.nz0 & 0xfu) ? 1u : 0u) | ((W.nz0 & 0xf0u) ? 2u : 0u) | ((W.nz0 & 0xf00u) ? 4u : 0u) | ((W.nz0 & 0xf000u) ? 8u : 0u);
        group([&](int w) { const int blk = 4 * w + g, pos = 15 - q; return ((W.nz0 >> blk) & 1u) && !(ac && pos == 0) ? abs((int)lvs[blk * 16 + pos]) : 0; }, 4, wm, 4, cb.r, ac ? 8 : 0, std::integral_constant<int, 32>{}, ac ? 1 : 0, ac ? 14 : 15, 0, 16);
    }
    if (W.ldc) group([&](int) { return lane < 16 ? abs((int)lvs[X264GPU_LV_LUMA_DC + 15 - lane]) : 0; }, 1, 1u, 4, cb.r, 24, std::integral_constant<int, 32>{}, 0, 15, 0, 16);
    if (W.nzdc) group([&](int) { return lane < 8 && ((W.nzdc >> (lane >> 2)) & 1u) ? abs((int)lvs[X264GPU_LV_CHROMA_DC + (lane >> 2) * 4 + 3 - (lane & 3)]) : 0; }, 1, 1u, 2, cb.r, 24, std::integral_constant<int, 55>{}, 0, 3, 48, 52);
    if (W.nzac) {
        const unsigned wm = ((W.nzac & 0xfu) ? 1u : 0u) | ((W.nzac & 0xf0u) ? 2u : 0u);
        group([&](int w) { const int blk = 4 * w + g, pos = 15 - q; return ((W.nzac >> blk) & 1u) && pos ? abs((int)lvs[X264GPU_LV_CHROMA_AC + blk * 16 + pos]) : 0; }, 2, wm, 4, cb.r, 16, std::integral_constant<int, 32>{}, 1, 14, 0, 16);
    }
    cb.f8 += f8;
}

// residual_block_cabac of a block of category CAT (0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC) up to its levels (cab_levels_all
// codes those for the whole macroblock): the significance map.  Coefficient i (scan order) in lane i of coef, zero elsewhere; the block holds a
// non-zero coefficient.  st = this lane's context variable of that category.
template <int CAT>
__device__ __forceinline__ void cab_block4(Cab &cb, int &st, uint32_t model, int lane, int coef)
{
    constexpr int n1 = CAT == 3 ? 3 : (CAT == 1 || CAT == 4) ? 14 : 15;
    constexpr int sig0 = CAT == 3 ? 48 : 0, last0 = CAT == 3 ? 52 : 16;
    const unsigned long long mask = __ballot(coef != 0);
    const int last = 63 - __builtin_clzll(mask);
    const bool is_s = lane >= sig0 && lane < sig0 + n1, is_l = lane >= last0 && lane < last0 + n1;
    const int p = is_s ? lane - sig0 : lane - last0;
    const int nzp = (int)((mask >> (p & 63)) & 1);
    cab_step(st, cb.f8v, model, (is_s && p <= last) || (is_l && nzp && p <= last), is_s ? nzp : p == last);
}

static __constant__ const unsigned long long c_cabac_pos8[24] = {
#include "cabac_masks8x8.inc"
};

// ... of an 8x8 luma block (64 coefficients, one per lane).  Positions share contexts here, so a lane walks the positions of ITS context:
// downwards as x264's size-only coder does (flags interleaved with the levels from the last coefficient down), upwards in bitstream order.
__device__ __forceinline__ void cab_block8(Cab &cb, int &st, uint32_t model, int lane, int coef, bool size)
{
    const unsigned long long mask = __ballot(coef != 0);
    const int last = 63 - __builtin_clzll(mask);
    const unsigned long long upto = last < 63 ? (2ull << last) - 1 : 0x7fffffffffffffffull;
    unsigned long long mine = 0;
    if (lane < 15) mine = c_cabac_pos8[lane] & upto;
    else if (lane >= 16 && lane < 25) mine = c_cabac_pos8[15 + lane - 16] & upto & mask;
    while (__ballot(mine != 0)) {
        const bool have = mine != 0;
        const int i = have ? (size ? 63 - __builtin_clzll(mine) : __builtin_ctzll(mine)) : 0;
        cab_step(st, cb.f8v, model, have, lane < 15 ? (int)((mask >> i) & 1) : i == last);
        mine &= ~(1ull << i);
    }
}

// what the coder needs to know about the macroblock and its neighbours (all wave-uniform)
struct CabIn {
    bool pslice, left, top, size;            // size: x264_macroblock_size_cabac (a candidate); else the finished macroblock in bitstream order
    int nref, t8mode;
    int type, part, t8, cbp_luma, cbp_chroma, i16mode, cmode, qp, last_qp, last_dqp;
    unsigned nnz;
    int ltype, ttype, lcbp_luma, tcbp_luma, lcbp_chroma, tcbp_chroma, lcmode, tcmode, lt8, tt8;
    unsigned lnnz, tnnz;
    unsigned long long lamvd, tamvd;         // |mvd| of the neighbours' 8x8 blocks: byte (block * 2 + component)
    // B slices: how the macroblock's 8x8 blocks are predicted (two bits each: 0 list 0, 1 list 1, 2 both, 3 direct), the motion of the lane's 8x8
    // block (lane >> 4) in both lists (per-lane values: block b's are read from lane 16 b), list 1's |mvd| neighbours
    bool bslice; int nref1; unsigned buse;
    int b_r0, b_x0, b_y0, b_r1, b_x1, b_y1;
    unsigned long long lamvd1, tamvd1;
    // RD refinement (x264 subme >= 8): the size of a PART of the macroblock (encoder/rdo.c partition_size_cabac, partition_i4x4 / _i8x8_size_cabac,
    // chroma_size_cabac) instead of the macroblock layer.  pm: 0 = the macroblock; 1 = inter part (P_L0 16x8 / 8x16 half, P_8x8 block): its mvd
    // (pm_dx, pm_dy against neighbours' |mvd| sums pm_sx, pm_sy), sub_mb_type for P_8x8, luma + chroma AC of its 8x8 blocks pm_b0 [, pm_b1];
    // 2 = Intra_4x4 block pm_b0: mode + block; 3 = Intra_8x8 block pm_b0: mode + the four cbp_luma bits + block; 4 = chroma of an intra
    // macroblock: mode + cbp_chroma + DC + AC.  Inside the macroblock the coded_block_flag neighbours read pm_nnzc — what x264's non_zero_count
    // cache holds, i.e. what the LAST encode of any kind left there (24 flags: luma blocks, chroma AC plane * 4 + block)
    int pm, pm_b0, pm_b1, pm_dx, pm_dy, pm_sx, pm_sy;
    int pm_lists, pm_dx1, pm_dy1, pm_sx1, pm_sy1;          // pm 1 in a B slice: the lists the part uses (bit 0 / 1), list 1's difference and sums
    unsigned pm_nnzc;
};

// mb_type of a B slice (Table 9-37 b): bin strings, first bin in bit 0
static __constant__ const uint8_t c_btype_len[24] = { 1, 3, 3, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 6, 6 };
static __constant__ const uint8_t c_btype_bits[24] = { 0, 1, 5, 3, 35, 19, 51, 11, 43, 27, 59, 31, 7, 71, 39, 103, 23, 87, 55, 119, 15, 79, 63, 47 };
__device__ __forceinline__ bool cab_is_skip(int t) { return t == X264GPU_MB_P_SKIP || t == X264GPU_MB_B_SKIP; }
__device__ __forceinline__ bool cab_is_intra(int t) { return t < X264GPU_MB_P_L0; }

__device__ __forceinline__ int cab_luma_cbf_of(int type, int cbp_luma, int t8, unsigned nnz, int bx, int by)
{
    if (type == X264GPU_MB_P_SKIP || type == X264GPU_MB_B_SKIP) return 0;
    if (!((cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1)) return 0;
    if (t8) return 1;
    return (nnz >> blkidx_of(bx, by)) & 1;
}

__device__ __forceinline__ void cab_mb_type_intra(Cab &cb, uint32_t model, int lane, const CabIn &in, int c0, int c1, int c2, int c3, int c4, int c5)
{
    if (in.type != X264GPU_MB_I16x16) { cab_bin(cb, model, lane, c0, 0); return; }
    cab_bin(cb, model, lane, c0, 1);
    if (in.size) cb.f8 += 7;                      // the terminate bin (not I_PCM) as x264's size macro prices it; it moves no context
    cab_bin(cb, model, lane, c1, in.cbp_luma != 0);
    if (!in.cbp_chroma) cab_bin(cb, model, lane, c2, 0);
    else { cab_bin(cb, model, lane, c2, 1); cab_bin(cb, model, lane, c3, in.cbp_chroma >> 1); }
    cab_bin(cb, model, lane, c4, in.i16mode >> 1);
    cab_bin(cb, model, lane, c5, in.i16mode & 1);
}

__device__ __forceinline__ void cab_mvd(Cab &cb, uint32_t model, int lane, int base, int sum, int val)
{
    const int a = abs(val), inc = (sum > 2) + (sum > 32);
    if (!a) { cab_bin(cb, model, lane, base + inc, 0); return; }
    cab_bin(cb, model, lane, base + inc, 1);
    for (int i = 1; i < min(a, 9); i++) cab_bin(cb, model, lane, base + min(i + 2, 6), 1);
    if (a < 9) cab_bin(cb, model, lane, base + min(a + 2, 6), 0); else cab_ue_bypass(cb, 3, a - 9);
    cab_bypass(cb);
}

// The macroblock layer.  S: the motion cache (neighbours + search results; partitions are cached into it as they are coded and it is
// restored before returning); lvs: the macroblock's levels (LDS); modes4 / modes8 / nmodes: intra modes (LDS).  Returns the |mvd| bytes of
// the macroblock's 8x8 blocks (zero for intra / skip) and the mb_qp_delta it sent through dqp_out.
template <class PROF>
__device__ __forceinline__ unsigned long long cab_mb(Cab &cb, uint32_t model, int lane, const CabIn &in, MeState &S, const int16_t *lvs,
                                                     const uint8_t *modes4, const uint8_t *modes8, const uint8_t *nmodes, int mbx, int sy, int &dqp_out,
                                                     unsigned long long &amvd1_out, const uint32_t *ctab, PROF &pf)
{
    pf.begin2(); pf.count(16);
    lane = relane(lane);
    amvd1_out = 0;
    unsigned long long amvd = 0;
    dqp_out = 0;
    const bool lavail = in.left, tavail = in.top;
    const bool intra = in.type < X264GPU_MB_P_L0;
    if (!in.pm) {
    if (in.bslice && !in.size) {
        cab_bin(cb, model, lane, 24 + (lavail && !cab_is_skip(in.ltype)) + (tavail && !cab_is_skip(in.ttype)), in.type == X264GPU_MB_B_SKIP);
        if (in.type == X264GPU_MB_B_SKIP) return 0;
    }
    if (in.pslice && !in.bslice && !in.size) {
        cab_bin(cb, model, lane, 11 + (lavail && in.ltype != X264GPU_MB_P_SKIP) + (tavail && in.ttype != X264GPU_MB_P_SKIP), in.type == X264GPU_MB_P_SKIP);
        if (in.type == X264GPU_MB_P_SKIP) return 0;
    }
    if (in.bslice) {
        // mb_type: Table 7-14 value -> bins; contexts 27 + {0..2} (neighbours that are neither B_Skip nor B_Direct_16x16), 27 + 3, 27 + 5 - b1, 27 + 5 ...
        const int ctx0 = (lavail && in.ltype != X264GPU_MB_B_SKIP && in.ltype != X264GPU_MB_B_DIRECT) + (tavail && in.ttype != X264GPU_MB_B_SKIP && in.ttype != X264GPU_MB_B_DIRECT);
        int value;
        if (intra) value = 23;
        else if (in.type == X264GPU_MB_B_DIRECT) value = 0;
        else if (in.part == D_8x8) value = 22;
        else if (in.part == D_16x16) value = 1 + (int)(in.buse & 3);
        else {
            const int u0 = (int)(in.buse & 3), u1 = (int)((in.part == D_16x8 ? in.buse >> 4 : in.buse >> 2) & 3);      // first / second partition: blocks 0 and 2 | 0 and 1
            const int pair = u0 == 0 ? (u1 == 0 ? 0 : u1 == 1 ? 2 : 4) : u0 == 1 ? (u1 == 0 ? 3 : u1 == 1 ? 1 : 5) : 6 + u1;
            value = 4 + 2 * pair + (in.part == D_8x16);
        }
        const int len = c_btype_len[value], bits = c_btype_bits[value];
        for (int i = 0; i < len; i++) cab_bin(cb, model, lane, i == 0 ? 27 + ctx0 : i == 1 ? 27 + 3 : i == 2 ? 27 + 5 - ((bits >> 1) & 1) : 27 + 5, (bits >> i) & 1);
        if (intra) cab_mb_type_intra(cb, model, lane, in, 32, 32 + 1, 32 + 2, 32 + 2, 32 + 3, 32 + 3);
    } else if (!in.pslice) {
        const int ctx = (lavail && in.ltype != X264GPU_MB_I4x4 && in.ltype != X264GPU_MB_I8x8) + (tavail && in.ttype != X264GPU_MB_I4x4 && in.ttype != X264GPU_MB_I8x8);
        cab_mb_type_intra(cb, model, lane, in, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
    } else if (intra) { cab_bin(cb, model, lane, 14, 1); cab_mb_type_intra(cb, model, lane, in, 17, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3); }
    else if (in.part == D_8x8) { cab_bin(cb, model, lane, 14, 0); cab_bin(cb, model, lane, 15, 0); cab_bin(cb, model, lane, 16, 1); }
    else {
        cab_bin(cb, model, lane, 14, 0);
        if (in.part == D_16x16) { cab_bin(cb, model, lane, 15, 0); cab_bin(cb, model, lane, 16, 0); }
        else { cab_bin(cb, model, lane, 15, 1); cab_bin(cb, model, lane, 17, in.part == D_16x8); }
    }
    }      // !in.pm
    const int t8ctx = 399 + (lavail && in.lt8) + (tavail && in.tt8);
    if (in.pm == 1) {
        // partition_size_cabac of a P part: its vector difference, the sub-macroblock type of a P_8x8 block
        if (in.pm_lists & 1) { cab_mvd(cb, model, lane, 40, in.pm_sx, in.pm_dx); cab_mvd(cb, model, lane, 47, in.pm_sy, in.pm_dy); }
        if (in.pm_lists & 2) { cab_mvd(cb, model, lane, 40, in.pm_sx1, in.pm_dx1); cab_mvd(cb, model, lane, 47, in.pm_sy1, in.pm_dy1); }
        if (in.part == D_8x8 && !in.bslice) cab_bin(cb, model, lane, 21, 1);
    } else if (intra) {
        if (in.type != X264GPU_MB_I16x16 && (!in.pm || in.pm == 2 || in.pm == 3)) {
            const bool i8 = in.type == X264GPU_MB_I8x8;
            if (in.t8mode && !in.pm) cab_bin(cb, model, lane, t8ctx, i8);
            // every lane its block's mode and predicted mode; the bins go out block by block
            int mode = 0, pm = 0;
            if (lane < 16) { const uint8_t *cur = i8 ? modes8 : modes4; mode = cur[lane]; pm = i4_pred_mode(nmodes, mbx, sy, lane, cur); }
            // prev_intra_pred_mode_flag (context 68) and the three rem_intra_pred_mode bins (context 69) are two chains: lane 0 and lane 1 walk
            // them side by side, block after block (contexts 68 and 69 share a dword of register a with 70 and 71, which nothing uses)
            const uint32_t w68 = __builtin_amdgcn_readlane(cb.a, 68 >> 2);
            int stp = lane == 0 ? (int)(w68 & 255) : (int)((w68 >> 8) & 255);
            for (int b = 0; b < 16; b += i8 ? 4 : 1) {
                if (in.pm && b != (i8 ? 4 * in.pm_b0 : in.pm_b0)) continue;
                int m = __builtin_amdgcn_readlane(mode, b);
                const int p = __builtin_amdgcn_readlane(pm, b);
                const bool same = m == p;
                if (m > p) m--;
                cab_step(stp, cb.f8v, model, lane == 0 || (lane == 1 && !same), lane == 0 ? same : m & 1);
                if (!same) { cab_step(stp, cb.f8v, model, lane == 1, (m >> 1) & 1); cab_step(stp, cb.f8v, model, lane == 1, m >> 2); }
            }
            {
                const uint32_t s68 = (uint32_t)__builtin_amdgcn_readlane(stp, 0), s69 = (uint32_t)__builtin_amdgcn_readlane(stp, 1);
                cb.a = lane == (68 >> 2) ? (cb.a & ~0xffffu) | s68 | (s69 << 8) : cb.a;
            }
        }
        if (!in.pm || in.pm == 4) {
        const int ctx = (lavail && in.ltype < X264GPU_MB_P_L0 && in.lcmode != 0) + (tavail && in.ttype < X264GPU_MB_P_L0 && in.tcmode != 0);
        if (!in.cmode) cab_bin(cb, model, lane, 64 + ctx, 0);
        else { cab_bin(cb, model, lane, 64 + ctx, 1); cab_bin(cb, model, lane, 64 + 3, in.cmode > 1); if (in.cmode > 1) cab_bin(cb, model, lane, 64 + 3, in.cmode > 2); }
        }
    } else if (in.bslice) {
        // ---- B: sub_mb_type, then every list-0 reference index, every list-1 one, list 0's vector differences, list 1's (7.3.5.1 / 7.3.5.2) ----
        if (in.type != X264GPU_MB_B_DIRECT) {
            const int part = in.part, np = part == D_16x16 ? 1 : part == D_8x8 ? 4 : 2;
            const int w8 = part == D_16x16 || part == D_16x8 ? 2 : 1, h8 = part == D_16x16 || part == D_8x16 ? 2 : 1;
            auto use_of = [&](int kp) { const int b8 = part == D_16x8 ? 2 * kp : kp; return (int)((in.buse >> (2 * b8)) & 3); };
            if (part == D_8x8)
                for (int kp = 0; kp < 4; kp++) {
                    const int u = use_of(kp);
                    if (u == 3) { cab_bin(cb, model, lane, 36, 0); continue; }
                    cab_bin(cb, model, lane, 36, 1);
                    if (u == 2) { cab_bin(cb, model, lane, 37, 1); cab_bin(cb, model, lane, 38, 0); cab_bin(cb, model, lane, 39, 0); cab_bin(cb, model, lane, 39, 0); }
                    else { cab_bin(cb, model, lane, 37, 0); cab_bin(cb, model, lane, 39, u == 1); }
                }
            const int sc0 = S.cref, sc1 = S.cmvx, sc2 = S.cmvy, sc3 = S.cdir;
            for (int l = 0; l < 2; l++) {
                const int go = 16 * l;
                if ((l ? in.nref1 : in.nref) <= 1) continue;
                if (lane == 5 + go || lane == 6 + go || lane == 9 + go || lane == 10 + go) S.cref = -2;
                for (int kp = 0; kp < np; kp++) {
                    const int x8 = part == D_8x16 ? kp : part == D_8x8 ? kp & 1 : 0, y8 = part == D_16x8 ? kp : part == D_8x8 ? kp >> 1 : 0;
                    const int u = use_of(kp), b8 = y8 * 2 + x8, g0 = (y8 + 1) * 4 + x8 + 1 + go;
                    const bool sends = u == 2 || u == l;
                    const int r = rl(l ? in.b_r1 : in.b_r0, b8 * 16);
                    if (sends) {
                        int ctx = (rl(S.cref, g0 - 1) > 0 && !rl(S.cdir, g0 - 1)) + 2 * (rl(S.cref, g0 - 4) > 0 && !rl(S.cdir, g0 - 4));
                        for (int q = r; q > 0; q--) { cab_bin(cb, model, lane, 54 + ctx, 1); ctx = (ctx >> 2) + 4; }
                        cab_bin(cb, model, lane, 54 + ctx, 0);
                    }
                    const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
                    S.cref = mine ? (sends ? r : u == 3 ? r : -1) : S.cref;
                    S.cdir = mine ? (u == 3) : S.cdir;
                }
            }
            unsigned long long amvd1 = 0;
            for (int l = 0; l < 2; l++) {
                const int go = 16 * l;
                if (lane == 5 + go || lane == 6 + go || lane == 9 + go || lane == 10 + go) S.cref = -2;
                unsigned long long am = 0;
                const unsigned long long nl = l ? in.lamvd1 : in.lamvd, nt = l ? in.tamvd1 : in.tamvd;
                for (int kp = 0; kp < np; kp++) {
                    const int x8 = part == D_8x16 ? kp : part == D_8x8 ? kp & 1 : 0, y8 = part == D_16x8 ? kp : part == D_8x8 ? kp >> 1 : 0;
                    const int u = use_of(kp), b8 = y8 * 2 + x8, g0 = (y8 + 1) * 4 + x8 + 1 + go;
                    const bool sends = u == 2 || u == l;
                    const int r = rl(l ? in.b_r1 : in.b_r0, b8 * 16), vx = rl(l ? in.b_x1 : in.b_x0, b8 * 16), vy = rl(l ? in.b_y1 : in.b_y0, b8 * 16);
                    if (sends) {
                        int px, py;
                        mb_predict_mv(S, part, x8, y8, w8, r, px, py, go);
                        for (int comp = 0; comp < 2; comp++) {
                            const int la = x8 > 0 ? (int)((am >> (8 * ((y8 * 2 + x8 - 1) * 2 + comp))) & 255) : lavail ? (int)((nl >> (8 * ((y8 * 2 + 1) * 2 + comp))) & 255) : 0;
                            const int ta = y8 > 0 ? (int)((am >> (8 * (((y8 - 1) * 2 + x8) * 2 + comp))) & 255) : tavail ? (int)((nt >> (8 * ((2 + x8) * 2 + comp))) & 255) : 0;
                            const int d = comp ? vy - py : vx - px;
                            cab_mvd(cb, model, lane, comp ? 47 : 40, la + ta, d);
                            const unsigned long long capped = (unsigned long long)min(abs(d), 66);
                            for (int yy = y8; yy < y8 + h8; yy++) for (int xx = x8; xx < x8 + w8; xx++) am |= capped << (8 * ((yy * 2 + xx) * 2 + comp));
                        }
                    }
                    const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
                    S.cref = mine ? (r >= 0 ? r : -1) : S.cref; S.cmvx = mine ? (r >= 0 ? vx : 0) : S.cmvx; S.cmvy = mine ? (r >= 0 ? vy : 0) : S.cmvy;
                }
                if (l) amvd1 = am; else amvd = am;
            }
            S.cref = sc0; S.cmvx = sc1; S.cmvy = sc2; S.cdir = sc3;
            amvd1_out = amvd1;
        }
    } else {
        const int part = in.part, np = part == D_16x16 ? 1 : part == D_8x8 ? 4 : 2;
        const int w8 = part == D_16x16 || part == D_16x8 ? 2 : 1, h8 = part == D_16x16 || part == D_8x16 ? 2 : 1;
        if (part == D_8x8) for (int kp = 0; kp < 4; kp++) cab_bin(cb, model, lane, 21, 1);          // sub_mb_type P_L0_8x8
        const int sc0 = S.cref, sc1 = S.cmvx, sc2 = S.cmvy;
        if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = -2;
        if (in.nref > 1)
            for (int kp = 0; kp < np; kp++) {
                const int x8 = part == D_8x16 ? kp : part == D_8x8 ? kp & 1 : 0, y8 = part == D_16x8 ? kp : part == D_8x8 ? kp >> 1 : 0;
                const int slot = part == D_16x16 ? ME_16 : part == D_16x8 ? ME_16x8 + kp : part == D_8x16 ? ME_8x16 + kp : ME_8 + kp;
                const int r = rl(S.ref, slot), g0 = (y8 + 1) * 4 + x8 + 1;
                // ref_idx: neighbours with a reference above 0 (skipped and intra macroblocks have none)
                int ctx = (rl(S.cref, g0 - 1) > 0) + 2 * (rl(S.cref, g0 - 4) > 0);
                for (int q = r; q > 0; q--) { cab_bin(cb, model, lane, 54 + ctx, 1); ctx = (ctx >> 2) + 4; }
                cab_bin(cb, model, lane, 54 + ctx, 0);
                const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
                S.cref = mine ? r : S.cref;
            }
        else if (lane == 5 || lane == 6 || lane == 9 || lane == 10) S.cref = 0;
        for (int kp = 0; kp < np; kp++) {
            const int x8 = part == D_8x16 ? kp : part == D_8x8 ? kp & 1 : 0, y8 = part == D_16x8 ? kp : part == D_8x8 ? kp >> 1 : 0;
            const int slot = part == D_16x16 ? ME_16 : part == D_16x8 ? ME_16x8 + kp : part == D_8x16 ? ME_8x16 + kp : ME_8 + kp;
            const int r = rl(S.ref, slot), vx = rl(S.mvx, slot), vy = rl(S.mvy, slot), g0 = (y8 + 1) * 4 + x8 + 1;
            int px, py;
            mb_predict_mv(S, part, x8, y8, w8, r, px, py);
            for (int comp = 0; comp < 2; comp++) {
                const int la = x8 > 0 ? (int)((amvd >> (8 * ((y8 * 2 + x8 - 1) * 2 + comp))) & 255) : lavail ? (int)((in.lamvd >> (8 * ((y8 * 2 + 1) * 2 + comp))) & 255) : 0;
                const int ta = y8 > 0 ? (int)((amvd >> (8 * (((y8 - 1) * 2 + x8) * 2 + comp))) & 255) : tavail ? (int)((in.tamvd >> (8 * ((2 + x8) * 2 + comp))) & 255) : 0;
                const int d = comp ? vy - py : vx - px;
                cab_mvd(cb, model, lane, comp ? 47 : 40, la + ta, d);
                const unsigned long long capped = (unsigned long long)min(abs(d), 66);
                for (int yy = y8; yy < y8 + h8; yy++) for (int xx = x8; xx < x8 + w8; xx++) amvd |= capped << (8 * ((yy * 2 + xx) * 2 + comp));
            }
            const bool mine = lane == g0 || (w8 == 2 && lane == g0 + 1) || (h8 == 2 && lane == g0 + 4) || (w8 == 2 && h8 == 2 && lane == g0 + 5);
            S.cmvx = mine ? vx : S.cmvx; S.cmvy = mine ? vy : S.cmvy;
        }
        S.cref = sc0; S.cmvx = sc1; S.cmvy = sc2;
    }
    if (in.pm ? in.pm == 3 : in.type != X264GPU_MB_I16x16)
        for (int b8 = 0; b8 < 4; b8++) {
            const int x = b8 & 1, y = b8 >> 1;
            const int a = x ? !((in.cbp_luma >> (b8 - 1)) & 1) : lavail ? !((in.lcbp_luma >> (b8 + 1)) & 1) : 0;
            const int b = y ? !((in.cbp_luma >> (b8 - 2)) & 1) : tavail ? !((in.tcbp_luma >> (b8 + 2)) & 1) : 0;
            cab_bin(cb, model, lane, 73 + a + 2 * b, (in.cbp_luma >> b8) & 1);
        }
    if (in.pm ? in.pm == 4 : in.type != X264GPU_MB_I16x16) {
        cab_bin(cb, model, lane, 77 + (lavail && in.lcbp_chroma) + 2 * (tavail && in.tcbp_chroma), in.cbp_chroma != 0);
        if (in.cbp_chroma) cab_bin(cb, model, lane, 77 + 4 + (lavail && in.lcbp_chroma == 2) + 2 * (tavail && in.tcbp_chroma == 2), in.cbp_chroma == 2);
    }
    if (!in.pm && !intra && in.t8mode && in.cbp_luma) cab_bin(cb, model, lane, t8ctx, in.t8);
    if (in.pm || in.cbp_luma || in.cbp_chroma || in.type == X264GPU_MB_I16x16) {
        const bool i16 = !in.pm && in.type == X264GPU_MB_I16x16;
        int dqp = in.qp - in.last_qp;
        // an I16x16 with nothing coded, DC included, never raises the quantiser (x264's qp_delta writers): it is sent as "no change"
        if (i16 && !in.cbp_luma && !in.cbp_chroma && !((in.nnz >> 24) & 1) && dqp > 0) dqp = 0;
        if (!in.pm) {
        int ctx = in.last_dqp != 0;
        if (dqp) {
            if (dqp < -26) dqp += 52; else if (dqp > 25) dqp -= 52;
            int val = dqp > 0 ? 2 * dqp - 1 : -2 * dqp;
            do { cab_bin(cb, model, lane, 60 + ctx, 1); ctx = 2 + (ctx >> 1); } while (--val);
        }
        cab_bin(cb, model, lane, 60 + ctx, 0);
        dqp_out = dqp;
        }
        pf.mark2(17);
        const int un = intra ? 1 : 0;
        // neighbour terms of the coded_block_flag contexts
        auto luma_in = [&](int bx, int by) { return in.pm ? (int)((in.pm_nnzc >> blkidx_of(bx, by)) & 1) : cab_luma_cbf_of(in.type, in.cbp_luma, in.t8, in.nnz, bx, by); };
        auto luma_inc = [&](int blk) {
            const int bx = z_bx(blk), by = z_by(blk);
            const int a = bx > 0 ? luma_in(bx - 1, by) : lavail ? cab_luma_cbf_of(in.ltype, in.lcbp_luma, in.lt8, in.lnnz, 3, by) : un;
            const int b = by > 0 ? luma_in(bx, by - 1) : tavail ? cab_luma_cbf_of(in.ttype, in.tcbp_luma, in.tt8, in.tnnz, bx, 3) : un;
            return a + 2 * b;
        };
        auto dc_inc = [&](int bit) {
            auto of = [&](bool avail, int type, int cbp_chroma, unsigned nnz) {
                if (!avail) return un;
                if (cab_is_skip(type)) return 0;
                if (bit == 24) return type == X264GPU_MB_I16x16 ? (int)((nnz >> 24) & 1) : 0;
                return cbp_chroma ? (int)((nnz >> bit) & 1) : 0;
            };
            return of(lavail, in.ltype, in.lcbp_chroma, in.lnnz) + 2 * of(tavail, in.ttype, in.tcbp_chroma, in.tnnz);
        };
        auto ac_inc = [&](int pl, int i) {
            const int bx = i & 1, by = i >> 1;
            auto of = [&](int type, int cbp_chroma, unsigned nnz, int x, int y) { return !cab_is_skip(type) && cbp_chroma == 2 ? (int)((nnz >> (16 + pl * 4 + y * 2 + x)) & 1) : 0; };
            auto inside = [&](int x, int y) { return in.pm ? (int)((in.pm_nnzc >> (16 + pl * 4 + y * 2 + x)) & 1) : of(in.type, in.cbp_chroma, in.nnz, x, y); };
            const int a = bx > 0 ? inside(0, by) : lavail ? of(in.ltype, in.lcbp_chroma, in.lnnz, 1, by) : un;
            const int b = by > 0 ? inside(bx, 0) : tavail ? of(in.ttype, in.tcbp_chroma, in.tnnz, bx, 1) : un;
            return a + 2 * b;
        };
        // residual: the category's context variables are this lane's byte of r (r8 for 8x8 blocks) for the duration of its blocks
        // (the levels of every block are coded at the end, all categories side by side: W collects which blocks hold coefficients)
        CabLv W = { -1, 0u, 0u, 0u, false };
        auto flush_levels = [&]() { pf.mark2(18); cab_levels_all(cb, model, lane, lvs, W, ctab, pf); pf.mark2(19); W.cat0 = -1; W.nz0 = 0; W.nzac = 0; W.nzdc = 0; W.ldc = false; };
        auto blocks = [&](auto cat_tag, int shift, int nblk, auto coef_of, auto inc_of, auto coded) {
            constexpr int CAT = decltype(cat_tag)::value;
            static_assert(CAT == 0 || CAT == 3, "the DC categories");
            (void)shift;
            for (int b = 0; b < nblk; b++) {
                if (!coded(b)) continue;
                const int coef = coef_of(b);
                const bool nz = __ballot(coef != 0) != 0;
                cab_bin(cb, model, lane, 85 + CAT * 4 + inc_of(b), nz);
                if (nz) { if (CAT == 0) W.ldc = true; else W.nzdc |= 1u << b; }          // (the block itself: cab_levels_all)
            }
        };
        auto always = [](int) { return true; };
        // The same for the many-block categories (luma 4x4, luma AC, chroma AC), faster: the coded_block_flags of the blocks are known up front
        // (nzbits: the macroblock's non-zero flags), and they share FOUR contexts — lane j takes context increment j and walks the blocks
        // that use it, in order, side by side with the other three; then only the blocks that hold coefficients are visited
        auto blocks_many = [&](auto cat_tag, int shift, int nblk, unsigned codedmask, unsigned nzbits, auto coef_of, auto inc_of) {
            constexpr int CAT = decltype(cat_tag)::value;
            const int b_lane = lane & 15;
            const bool mine_blk = lane < nblk && ((codedmask >> b_lane) & 1);
            const int inc = inc_of(b_lane);
            const unsigned long long nzm = (unsigned long long)(nzbits & codedmask);
            unsigned long long seq = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { const unsigned long long m = __ballot(mine_blk && inc == j); seq = lane == j ? m : seq; }
            const int cbase = 85 + CAT * 4;
            int stc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { const int c = cbase + j; const uint32_t w = __builtin_amdgcn_readlane(cb.a, c >> 2); stc = lane == j ? (int)((w >> ((c & 3) * 8)) & 255) : stc; }
            while (__ballot(seq != 0)) {
                const bool have = seq != 0;
                const int b = have ? __builtin_ctzll(seq) : 0;
                cab_step(stc, cb.f8v, model, have, (int)((nzm >> b) & 1));
                seq &= seq - 1;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = cbase + j, li = c >> 2, sh = (c & 3) * 8;
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(stc, j);
                cb.a = lane == li ? (cb.a & ~(255u << sh)) | (v << sh) : cb.a;
            }
            (void)coef_of; (void)shift;          // (the blocks themselves — significance maps and levels — are coded by cab_levels_all at the end)
            if (CAT == 4) W.nzac |= (unsigned)nzm; else { W.cat0 = CAT; W.nz0 |= (unsigned)nzm; }
        };
        if (i16) {
            blocks(std::integral_constant<int, 0>{}, 24, 1, [&](int) { return lane < 16 ? (int)lvs[X264GPU_LV_LUMA_DC + lane] : 0; }, [&](int) { return dc_inc(24); }, always);
            if (in.cbp_luma) blocks_many(std::integral_constant<int, 1>{}, 8, 16, 0xffffu, in.nnz & 0xffffu, [&](int b) { return lane < 15 ? (int)lvs[b * 16 + 1 + lane] : 0; }, luma_inc);
        } else if (in.pm == 4) {
        } else if (in.t8) {
            // (part mode: only the part's 8x8 blocks — pm_b1 < 0: one block)
            const int pmask = in.pm ? (1 << in.pm_b0) | (in.pm_b1 >= 0 ? 1 << in.pm_b1 : 0) : 15;
            int st = cb.r8 & 255;
            for (int i8 = 0; i8 < 4; i8++)
                if (((in.cbp_luma & pmask) >> i8) & 1) cab_block8(cb, st, model, lane, (int)lvs[(i8 * 4 + (lane & 3)) * 16 + (lane >> 2)], in.size);
            cb.r8 = (uint32_t)st;
            W.cat0 = 5; W.nz0 = (unsigned)(in.cbp_luma & pmask);
        } else {
            unsigned coded = ((in.cbp_luma & 1) ? 0x000fu : 0) | ((in.cbp_luma & 2) ? 0x00f0u : 0) | ((in.cbp_luma & 4) ? 0x0f00u : 0) | ((in.cbp_luma & 8) ? 0xf000u : 0);
            if (in.pm == 1) coded &= (0xfu << (4 * in.pm_b0)) | (in.pm_b1 >= 0 ? 0xfu << (4 * in.pm_b1) : 0u);
            if (in.pm == 2) coded = 1u << in.pm_b0;          // an Intra_4x4 block always sends its coded_block_flag
            blocks_many(std::integral_constant<int, 2>{}, 0, 16, coded, in.nnz & 0xffffu, [&](int b) { return lane < 16 ? (int)lvs[b * 16 + lane] : 0; }, luma_inc);
        }
        if (in.pm == 1) {
            // the chroma AC blocks under the part's 8x8 blocks, block by block (U then V of the first, then of the second)
            for (int t = 0; t < 2; t++) {
                const int b8 = t ? in.pm_b1 : in.pm_b0;
                if (b8 < 0) continue;
                blocks_many(std::integral_constant<int, 4>{}, 16, 8, (1u << b8) | (16u << b8), (in.nnz >> 16) & 0xffu, [&](int k) { return lane < 15 ? (int)lvs[X264GPU_LV_CHROMA_AC + k * 16 + 1 + lane] : 0; }, [&](int k) { return ac_inc((k >> 2) & 1, k & 3); });
                flush_levels();          // (U then V of this 8x8 block before the next one's: the planes share their contexts, the walk goes plane by plane)
            }
        } else if (in.cbp_chroma && (!in.pm || in.pm == 4)) {
            blocks(std::integral_constant<int, 3>{}, 24, 2, [&](int pl) { return lane < 4 ? (int)lvs[X264GPU_LV_CHROMA_DC + pl * 4 + lane] : 0; }, [&](int pl) { return dc_inc(25 + pl); }, always);
            if (in.cbp_chroma == 2)
                blocks_many(std::integral_constant<int, 4>{}, 16, 8, 0xffu, (in.nnz >> 16) & 0xffu, [&](int k) { return lane < 15 ? (int)lvs[X264GPU_LV_CHROMA_AC + k * 16 + 1 + lane] : 0; }, [&](int k) { return ac_inc((k >> 2) & 1, k & 3); });
        }
        flush_levels();
    }
    return amvd;
}

}  // namespace x264gpu

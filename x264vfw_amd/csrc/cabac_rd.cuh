// cabac_rd.cuh — CABAC as x264's rate-distortion code needs it on the device (k_mb.cuh, RD instantiations of CABAC sessions).
//
// x264 ([x264-upstream] encoder/cabac.c + encoder/rdo.c behind x264_encoder_encode, reference call site codec.c:1693) keeps one set of
// context variables per slice, moved on by the entropy coding of every finished macroblock, and prices a candidate by running the
// macroblock's syntax over a COPY of them with the arithmetic coder replaced by a counter in 1/256 bit (entropy[state ^ bin], bypass
// bin = 256, the I16x16 terminate bin = 7).  The bitstream itself is still written by the host (host/cabac.cpp); what lives here is
//   * the 460 context variables of the slice a wavefront codes: ONE BYTE EACH, FOUR TO A DWORD, IN TWO VGPRs (context c sits in byte c & 3
//     of lane (c >> 2) & 63 of register c >> 8) — copying them for a candidate is two moves, and the coder below, whose control flow is
//     wave-uniform, reads and writes them with v_readlane / a lane select;
//   * a third register with the probability model: lane s = cost of the MPS | cost of the LPS << 9 | state after an LPS << 20;
//   * cab_mb(): the macroblock layer's bins, either in bitstream order ("evolve": what the finished macroblock leaves behind, skip flag
//     included) or as x264_macroblock_size_cabac walks them ("size": no skip flag, residual blocks from the last coefficient down with
//     flags and levels interleaved — the order matters for 8x8 blocks, whose positions share contexts).
// Mirrors oracle/cabac_rd.cpp bin for bin; the initialisation values and state transitions are the host coder's tables.
#pragma once
#include "enc_common.cuh"
#define CABAC_TABLE static __constant__ const
#define CABAC_NAMESPACE x264gpu_cabac
#include "../host/cabac_tables.hpp"
#undef CABAC_TABLE
#undef CABAC_NAMESPACE

namespace x264gpu {

static __constant__ const uint16_t c_cabac_entropy[128] = {
#include "cabac_entropy.inc"
};

struct Cab { uint32_t a, b; int f8; };

__device__ __forceinline__ uint32_t cab_model(int lane)
{
    return (uint32_t)c_cabac_entropy[2 * lane] | ((uint32_t)c_cabac_entropy[2 * lane + 1] << 9) | ((uint32_t)x264gpu_cabac::cabac_trans_lps[lane] << 20);
}

// 9.3.1.1: context variables from the slice quantiser (cabac_init_idc 0)
__device__ __forceinline__ void cab_init(Cab &cb, int lane, bool pslice, int qp)
{
    namespace T = x264gpu_cabac;
    qp = min(max(qp, 0), 51);
    uint32_t w[2] = { 0, 0 };
    for (int r = 0; r < 2; r++)
        for (int j = 0; j < 4; j++) {
            const int ctx = r * 256 + lane * 4 + j;
            int m = 0, n = 0;
            if (ctx < 276) { const T::CabacInitRow row = T::cabac_init_0_275[ctx]; m = pslice ? row.mp : row.mi; n = pslice ? row.np : row.ni; }
            else if (ctx >= 399 && ctx <= 435) { const T::CabacInitRow row = T::cabac_init_399_435[ctx - 399]; m = pslice ? row.mp : row.mi; n = pslice ? row.np : row.ni; }
            const int pre = min(max(((m * qp) >> 4) + n, 1), 126);
            const int st = pre <= 63 ? (63 - pre) << 1 : ((pre - 64) << 1) | 1;
            w[r] |= (uint32_t)st << (8 * j);
        }
    cb.a = w[0]; cb.b = w[1]; cb.f8 = 0;
}

__device__ __forceinline__ void cab_bin(Cab &cb, uint32_t model, int lane, int ctx, int bin)
{
    ctx = __builtin_amdgcn_readfirstlane(ctx); bin = __builtin_amdgcn_readfirstlane(bin);
    const int li = (ctx >> 2) & 63, sh = (ctx & 3) * 8;
    const bool hi = ctx >= 256;
    const uint32_t w = hi ? __builtin_amdgcn_readlane(cb.b, li) : __builtin_amdgcn_readlane(cb.a, li);
    const int st = (w >> sh) & 255, sg = st >> 1, mps = st & 1;
    const uint32_t t = __builtin_amdgcn_readlane(model, sg);
    const bool lps = (mps ^ bin) != 0;
    cb.f8 += lps ? (t >> 9) & 0x7ff : t & 0x1ff;
    const int ns = lps ? (int)(t >> 20) : min(sg + 1, 62);
    const int nm = lps && sg == 0 ? mps ^ 1 : mps;
    const uint32_t nw = (w & ~(255u << sh)) | ((uint32_t)((ns << 1) | nm) << sh);
    if (hi) cb.b = lane == li ? nw : cb.b; else cb.a = lane == li ? nw : cb.a;
}
__device__ __forceinline__ void cab_bypass(Cab &cb, int n = 1) { cb.f8 += 256 * n; }
__device__ __forceinline__ void cab_ue_bypass(Cab &cb, int k, int v)
{
    int n = 0;
    while (v >= (1 << k)) { n++; v -= 1 << k; k++; }
    cb.f8 += 256 * (n + 1 + k);
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
};

__device__ __forceinline__ int cab_luma_cbf_of(int type, int cbp_luma, int t8, unsigned nnz, int bx, int by)
{
    if (type == X264GPU_MB_P_SKIP) return 0;
    if (!((cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1)) return 0;
    if (t8) return 1;
    return (nnz >> blkidx_of(bx, by)) & 1;
}

// one level: coeff_abs_level_minus1 prefix / suffix + sign, x264's node contexts
__device__ __forceinline__ void cab_level(Cab &cb, uint32_t model, int lane, int abs_off, int v, int &node)
{
    const int a = abs(v);
    const int c1 = node < 4 ? node + 1 : 0, cg = node < 4 ? 5 : min(node + 2, 9);
    if (a > 1) {
        cab_bin(cb, model, lane, abs_off + c1, 1);
        for (int i = min(a, 15) - 2; i > 0; i--) cab_bin(cb, model, lane, abs_off + cg, 1);
        if (a < 15) cab_bin(cb, model, lane, abs_off + cg, 0); else cab_ue_bypass(cb, 0, a - 15);
        node = node < 4 ? 4 : min(node + 1, 7);
    } else { cab_bin(cb, model, lane, abs_off + c1, 0); node = node < 3 ? node + 1 : node == 3 ? 3 : node; }
    cab_bypass(cb);
}

// residual_block_cabac of the block whose coefficient i (scan order) sits in lane i of coef (zero beyond the block).  cat: 0 luma DC,
// 1 luma AC (15), 2 luma 4x4, 3 chroma DC (4), 4 chroma AC (15), 5 luma 8x8.  The block is known to hold a non-zero coefficient.
__device__ __forceinline__ void cab_residual(Cab &cb, uint32_t model, int lane, int coef, int cat, bool size)
{
    namespace T = x264gpu_cabac;
    const int sig_off = cat == 0 ? 105 : cat == 1 ? 120 : cat == 2 ? 134 : cat == 3 ? 149 : cat == 4 ? 152 : 402;
    const int last_off = cat == 0 ? 166 : cat == 1 ? 181 : cat == 2 ? 195 : cat == 3 ? 210 : cat == 4 ? 213 : 417;
    const int abs_off = cat == 0 ? 227 : cat == 1 ? 237 : cat == 2 ? 247 : cat == 3 ? 257 : cat == 4 ? 266 : 426;
    const int n1 = cat == 3 ? 3 : cat == 5 ? 63 : (cat == 1 || cat == 4) ? 14 : 15;
    const unsigned long long mask = __ballot(coef != 0);
    const int last = 63 - __builtin_clzll(mask);
    int node = 0;
    if (size) {
        if (last != n1) {
            cab_bin(cb, model, lane, sig_off + (cat == 5 ? T::cabac_sig8x8[last] : last), 1);
            cab_bin(cb, model, lane, last_off + (cat == 5 ? T::cabac_last8x8[last] : last), 1);
        }
        cab_level(cb, model, lane, abs_off, __builtin_amdgcn_readlane(coef, last), node);
        for (int i = last - 1; i >= 0; i--) {
            const int so = sig_off + (cat == 5 ? T::cabac_sig8x8[i] : i);
            if ((mask >> i) & 1) {
                cab_bin(cb, model, lane, so, 1);
                cab_bin(cb, model, lane, last_off + (cat == 5 ? T::cabac_last8x8[i] : i), 0);
                cab_level(cb, model, lane, abs_off, __builtin_amdgcn_readlane(coef, i), node);
            } else cab_bin(cb, model, lane, so, 0);
        }
    } else {
        for (int i = 0; i < last; i++) {
            const int nz = (mask >> i) & 1;
            cab_bin(cb, model, lane, sig_off + (cat == 5 ? T::cabac_sig8x8[i] : i), nz);
            if (nz) cab_bin(cb, model, lane, last_off + (cat == 5 ? T::cabac_last8x8[i] : i), 0);
        }
        if (last != n1) {
            cab_bin(cb, model, lane, sig_off + (cat == 5 ? T::cabac_sig8x8[last] : last), 1);
            cab_bin(cb, model, lane, last_off + (cat == 5 ? T::cabac_last8x8[last] : last), 1);
        }
        for (int i = last; i >= 0; i--) if ((mask >> i) & 1) cab_level(cb, model, lane, abs_off, __builtin_amdgcn_readlane(coef, i), node);
    }
}

// coded_block_flag + the block
__device__ __forceinline__ void cab_block_cbf(Cab &cb, uint32_t model, int lane, int coef, int cat, int inc, bool size)
{
    const bool nz = __ballot(coef != 0) != 0;
    cab_bin(cb, model, lane, 85 + cat * 4 + inc, nz);
    if (nz) cab_residual(cb, model, lane, coef, cat, size);
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
__device__ __forceinline__ unsigned long long cab_mb(Cab &cb, uint32_t model, int lane, const CabIn &in, MeState &S, const int16_t *lvs,
                                                     const uint8_t *modes4, const uint8_t *modes8, const uint8_t *nmodes, int mbx, int sy, int &dqp_out)
{
    unsigned long long amvd = 0;
    dqp_out = 0;
    const bool lavail = in.left, tavail = in.top;
    if (in.pslice && !in.size) {
        cab_bin(cb, model, lane, 11 + (lavail && in.ltype != X264GPU_MB_P_SKIP) + (tavail && in.ttype != X264GPU_MB_P_SKIP), in.type == X264GPU_MB_P_SKIP);
        if (in.type == X264GPU_MB_P_SKIP) return 0;
    }
    const bool intra = in.type < X264GPU_MB_P_L0;
    if (!in.pslice) {
        const int ctx = (lavail && in.ltype != X264GPU_MB_I4x4 && in.ltype != X264GPU_MB_I8x8) + (tavail && in.ttype != X264GPU_MB_I4x4 && in.ttype != X264GPU_MB_I8x8);
        cab_mb_type_intra(cb, model, lane, in, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
    } else if (intra) { cab_bin(cb, model, lane, 14, 1); cab_mb_type_intra(cb, model, lane, in, 17, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3); }
    else if (in.part == D_8x8) { cab_bin(cb, model, lane, 14, 0); cab_bin(cb, model, lane, 15, 0); cab_bin(cb, model, lane, 16, 1); }
    else {
        cab_bin(cb, model, lane, 14, 0);
        if (in.part == D_16x16) { cab_bin(cb, model, lane, 15, 0); cab_bin(cb, model, lane, 16, 0); }
        else { cab_bin(cb, model, lane, 15, 1); cab_bin(cb, model, lane, 17, in.part == D_16x8); }
    }
    const int t8ctx = 399 + (lavail && in.lt8) + (tavail && in.tt8);
    if (intra) {
        if (in.type != X264GPU_MB_I16x16) {
            const bool i8 = in.type == X264GPU_MB_I8x8;
            if (in.t8mode) cab_bin(cb, model, lane, t8ctx, i8);
            // every lane its block's mode and predicted mode; the bins go out block by block
            int mode = 0, pm = 0;
            if (lane < 16) { const uint8_t *cur = i8 ? modes8 : modes4; mode = cur[lane]; pm = i4_pred_mode(nmodes, mbx, sy, lane, cur); }
            for (int b = 0; b < 16; b += i8 ? 4 : 1) {
                int m = __builtin_amdgcn_readlane(mode, b);
                const int p = __builtin_amdgcn_readlane(pm, b);
                if (m == p) cab_bin(cb, model, lane, 68, 1);
                else {
                    cab_bin(cb, model, lane, 68, 0);
                    if (m > p) m--;
                    cab_bin(cb, model, lane, 69, m & 1); cab_bin(cb, model, lane, 69, (m >> 1) & 1); cab_bin(cb, model, lane, 69, m >> 2);
                }
            }
        }
        const int ctx = (lavail && in.ltype < X264GPU_MB_P_L0 && in.lcmode != 0) + (tavail && in.ttype < X264GPU_MB_P_L0 && in.tcmode != 0);
        if (!in.cmode) cab_bin(cb, model, lane, 64 + ctx, 0);
        else { cab_bin(cb, model, lane, 64 + ctx, 1); cab_bin(cb, model, lane, 64 + 3, in.cmode > 1); if (in.cmode > 1) cab_bin(cb, model, lane, 64 + 3, in.cmode > 2); }
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
    if (in.type != X264GPU_MB_I16x16) {
        for (int b8 = 0; b8 < 4; b8++) {
            const int x = b8 & 1, y = b8 >> 1;
            const int a = x ? !((in.cbp_luma >> (b8 - 1)) & 1) : lavail ? !((in.lcbp_luma >> (b8 + 1)) & 1) : 0;
            const int b = y ? !((in.cbp_luma >> (b8 - 2)) & 1) : tavail ? !((in.tcbp_luma >> (b8 + 2)) & 1) : 0;
            cab_bin(cb, model, lane, 73 + a + 2 * b, (in.cbp_luma >> b8) & 1);
        }
        cab_bin(cb, model, lane, 77 + (lavail && in.lcbp_chroma) + 2 * (tavail && in.tcbp_chroma), in.cbp_chroma != 0);
        if (in.cbp_chroma) cab_bin(cb, model, lane, 77 + 4 + (lavail && in.lcbp_chroma == 2) + 2 * (tavail && in.tcbp_chroma == 2), in.cbp_chroma == 2);
    }
    if (!intra && in.t8mode && in.cbp_luma) cab_bin(cb, model, lane, t8ctx, in.t8);
    if (in.cbp_luma || in.cbp_chroma || in.type == X264GPU_MB_I16x16) {
        const bool i16 = in.type == X264GPU_MB_I16x16;
        int dqp = in.qp - in.last_qp;
        // an I16x16 with nothing coded, DC included, never raises the quantiser (x264's qp_delta writers): it is sent as "no change"
        if (i16 && !in.cbp_luma && !in.cbp_chroma && !((in.nnz >> 24) & 1) && dqp > 0) dqp = 0;
        int ctx = in.last_dqp != 0;
        if (dqp) {
            if (dqp < -26) dqp += 52; else if (dqp > 25) dqp -= 52;
            int val = dqp > 0 ? 2 * dqp - 1 : -2 * dqp;
            do { cab_bin(cb, model, lane, 60 + ctx, 1); ctx = 2 + (ctx >> 1); } while (--val);
        }
        cab_bin(cb, model, lane, 60 + ctx, 0);
        dqp_out = dqp;
        const int un = intra ? 1 : 0;
        // neighbour terms of the coded_block_flag contexts
        auto luma_inc = [&](int blk) {
            const int bx = z_bx(blk), by = z_by(blk);
            const int a = bx > 0 ? cab_luma_cbf_of(in.type, in.cbp_luma, in.t8, in.nnz, bx - 1, by) : lavail ? cab_luma_cbf_of(in.ltype, in.lcbp_luma, in.lt8, in.lnnz, 3, by) : un;
            const int b = by > 0 ? cab_luma_cbf_of(in.type, in.cbp_luma, in.t8, in.nnz, bx, by - 1) : tavail ? cab_luma_cbf_of(in.ttype, in.tcbp_luma, in.tt8, in.tnnz, bx, 3) : un;
            return a + 2 * b;
        };
        auto dc_inc = [&](int bit) {
            auto of = [&](bool avail, int type, int cbp_chroma, unsigned nnz) {
                if (!avail) return un;
                if (type == X264GPU_MB_P_SKIP) return 0;
                if (bit == 24) return type == X264GPU_MB_I16x16 ? (int)((nnz >> 24) & 1) : 0;
                return cbp_chroma ? (int)((nnz >> bit) & 1) : 0;
            };
            return of(lavail, in.ltype, in.lcbp_chroma, in.lnnz) + 2 * of(tavail, in.ttype, in.tcbp_chroma, in.tnnz);
        };
        auto ac_inc = [&](int pl, int i) {
            const int bx = i & 1, by = i >> 1;
            auto of = [&](int type, int cbp_chroma, unsigned nnz, int x, int y) { return type != X264GPU_MB_P_SKIP && cbp_chroma == 2 ? (int)((nnz >> (16 + pl * 4 + y * 2 + x)) & 1) : 0; };
            const int a = bx > 0 ? of(in.type, in.cbp_chroma, in.nnz, 0, by) : lavail ? of(in.ltype, in.lcbp_chroma, in.lnnz, 1, by) : un;
            const int b = by > 0 ? of(in.type, in.cbp_chroma, in.nnz, bx, 0) : tavail ? of(in.ttype, in.tcbp_chroma, in.tnnz, bx, 1) : un;
            return a + 2 * b;
        };
        if (i16) {
            cab_block_cbf(cb, model, lane, lane < 16 ? (int)lvs[X264GPU_LV_LUMA_DC + lane] : 0, 0, dc_inc(24), in.size);
            if (in.cbp_luma) for (int b = 0; b < 16; b++) cab_block_cbf(cb, model, lane, lane < 15 ? (int)lvs[b * 16 + 1 + lane] : 0, 1, luma_inc(b), in.size);
        } else if (in.t8) {
            for (int i8 = 0; i8 < 4; i8++)
                if ((in.cbp_luma >> i8) & 1) cab_residual(cb, model, lane, (int)lvs[(i8 * 4 + (lane & 3)) * 16 + (lane >> 2)], 5, in.size);
        } else {
            for (int b = 0; b < 16; b++) if ((in.cbp_luma >> (b >> 2)) & 1) cab_block_cbf(cb, model, lane, lane < 16 ? (int)lvs[b * 16 + lane] : 0, 2, luma_inc(b), in.size);
        }
        if (in.cbp_chroma) {
            for (int pl = 0; pl < 2; pl++) cab_block_cbf(cb, model, lane, lane < 4 ? (int)lvs[X264GPU_LV_CHROMA_DC + pl * 4 + lane] : 0, 3, dc_inc(25 + pl), in.size);
            if (in.cbp_chroma == 2)
                for (int pl = 0; pl < 2; pl++)
                    for (int i = 0; i < 4; i++) cab_block_cbf(cb, model, lane, lane < 15 ? (int)lvs[X264GPU_LV_CHROMA_AC + (pl * 4 + i) * 16 + 1 + lane] : 0, 4, ac_inc(pl, i), in.size);
        }
    }
    return amvd;
}

}  // namespace x264gpu

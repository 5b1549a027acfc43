// dsp8.hip.h — the 8x8 luma transform path (High profile, A6-A8: sub8x8_dct8, quant_8x8, scan_8x8,
// decimate_score64, dequant_8x8, add8x8_idct8) and SA8D for gfx950.
//
// "R8 layout": lane = (8x8 block, row) — 8 lanes own one 8x8 block, each lane holds one row of 8 samples in
// registers; lanes 0..31 of a wavefront cover the four 8x8 blocks of a macroblock (lanes 32..63 mirror them).
// Row transforms are in-lane 8-point butterflies; column transforms run after an 8x8 transpose across the 8
// lanes (DPP quad_perm for lane^1 / lane^2, a cross-lane shuffle for lane^4).  Quantiser tables have six
// position classes (normAdjust8x8).  Restates oracle/dct.c + oracle/quant.c bit-exactly.
#pragma once
#include "dsp.hip.h"

namespace x264gpu {

struct Q8 { int mf[6], bias[6], dq[6], qp; };

// 8-sample rows without pointer arithmetic on the register arrays (keeps them out of promoted-alloca LDS)
__device__ __forceinline__ void unpack8(uint32_t lo, uint32_t hi, int v[8])
{
    v[0] = lo & 0xff; v[1] = (lo >> 8) & 0xff; v[2] = (lo >> 16) & 0xff; v[3] = lo >> 24;
    v[4] = hi & 0xff; v[5] = (hi >> 8) & 0xff; v[6] = (hi >> 16) & 0xff; v[7] = hi >> 24;
}
__device__ __forceinline__ uint32_t pack4_clip8lo(const int v[8])
{
    return (uint32_t)clip_u8(v[0]) | ((uint32_t)clip_u8(v[1]) << 8) | ((uint32_t)clip_u8(v[2]) << 16) | ((uint32_t)clip_u8(v[3]) << 24);
}
__device__ __forceinline__ uint32_t pack4_clip8hi(const int v[8])
{
    return (uint32_t)clip_u8(v[4]) | ((uint32_t)clip_u8(v[5]) << 8) | ((uint32_t)clip_u8(v[6]) << 16) | ((uint32_t)clip_u8(v[7]) << 24);
}

__device__ __forceinline__ void fwd8_1d(int s[8])
{
    const int s07 = s[0] + s[7], s16 = s[1] + s[6], s25 = s[2] + s[5], s34 = s[3] + s[4];
    const int d07 = s[0] - s[7], d16 = s[1] - s[6], d25 = s[2] - s[5], d34 = s[3] - s[4];
    const int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    const int a4 = d16 + d25 + (d07 + (d07 >> 1));
    const int a5 = d07 - d34 - (d25 + (d25 >> 1));
    const int a6 = d07 + d34 - (d16 + (d16 >> 1));
    const int a7 = d16 - d25 + (d34 + (d34 >> 1));
    s[0] = a0 + a1; s[1] = a4 + (a7 >> 2); s[2] = a2 + (a3 >> 1); s[3] = a5 + (a6 >> 2);
    s[4] = a0 - a1; s[5] = a6 - (a5 >> 2); s[6] = (a2 >> 1) - a3; s[7] = (a4 >> 2) - a7;
}
__device__ __forceinline__ void inv8_1d(int s[8])
{
    const int a0 = s[0] + s[4], a2 = s[0] - s[4], a4 = (s[2] >> 1) - s[6], a6 = (s[6] >> 1) + s[2];
    const int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    const int a1 = -s[3] + s[5] - s[7] - (s[7] >> 1);
    const int a3 = s[1] + s[7] - s[3] - (s[3] >> 1);
    const int a5 = -s[1] + s[7] + s[5] + (s[5] >> 1);
    const int a7 = s[3] + s[5] + s[1] + (s[1] >> 1);
    const int b1 = (a7 >> 2) + a1, b3 = a3 + (a5 >> 2), b5 = (a3 >> 2) - a5, b7 = a7 - (a1 >> 2);
    s[0] = b0 + b7; s[1] = b2 + b5; s[2] = b4 + b3; s[3] = b6 + b1;
    s[4] = b6 - b1; s[5] = b4 - b3; s[6] = b2 - b5; s[7] = b0 - b7;
}
// value of lane^4 (exchange between the two quads of an 8-lane group): row_shl:4 feeds lanes 0-3 / 8-11 of each DPP
// row, row_shr:4 feeds lanes 4-7 / 12-15 (bank masks select which quads a DPP move writes)
__device__ __forceinline__ int xor4(int v)
{
    int r = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xf, 0x5, false);      // row_shl:4 -> lane i takes lane i+4
    return __builtin_amdgcn_update_dpp(r, v, 0x114, 0xf, 0xa, false);       // row_shr:4 -> lane i takes lane i-4
}
// (lane r, reg c) -> (lane c, reg r) within each group of 8 lanes: three exchange stages.  The stage stride is a
// template parameter so every register index is a compile-time constant (a runtime-indexed v[] would be demoted to
// an LDS-backed array by the compiler, which is what the first version of this routine silently cost).
template <int S>
__device__ __forceinline__ void transpose8_stage(int v[8], int lane)
{
    const bool hi = lane & S;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (r & S) continue;
        const int lo_v = v[r], hi_v = v[r | S];                 // values first: `c ? v[a] : v[b]` is an lvalue select (a pointer select)
        const int send = hi ? lo_v : hi_v;
        const int recv = S == 1 ? dpp<DPP_XOR1>(send) : S == 2 ? dpp<DPP_XOR2>(send) : xor4(send);
        v[r] = hi ? recv : lo_v;
        v[r | S] = hi ? hi_v : recv;
    }
}
__device__ __forceinline__ void transpose8(int v[8], int lane)
{
    transpose8_stage<1>(v, lane);
    transpose8_stage<2>(v, lane);
    transpose8_stage<4>(v, lane);
}
__device__ __forceinline__ int class8(int r, int c)
{
    // normAdjust8x8 class of position (r,c): [r&3][c&3] -> { {0,3,4,3}, {3,1,5,1}, {4,5,2,5}, {3,1,5,1} }
    const unsigned long long tab = 0x1513525415133430ull;   // nibble (r&3)*4 + (c&3)
    return (int)((tab >> (((r & 3) * 4 + (c & 3)) * 4)) & 15);
}
// this lane's quantiser constants for columns c&3 of its row
__device__ __forceinline__ void q8_row(const Q8 &q, int row, int mf[4], int bias[4], int dq[4])
{
#pragma unroll
    for (int c = 0; c < 4; c++) { const int k = class8(row, c); mf[c] = q.mf[k]; bias[c] = q.bias[k]; dq[c] = q.dq[k]; }
}

// frame zigzag of the 8x8 block: scan index of raster position r*8+c
static __constant__ uint8_t c_zigzag8_inv[64] = {
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63 };

// OR over the 8 lanes of an R8 group
__device__ __forceinline__ unsigned group8_or(unsigned v)
{
    v |= (unsigned)dpp<DPP_XOR1>((int)v);
    v |= (unsigned)dpp<DPP_XOR2>((int)v);
    v |= (unsigned)xor4((int)v);
    return v;
}
// decimate_score64 from the scan-order nonzero mask (all levels known to be +-1): table {3 x4, 2 x8, 1 x20, 0..}
// over the zero run below each coefficient.  Saturates at >= 6 (callers only compare against 4 and 6).
__device__ __forceinline__ int decimate64_from_mask(unsigned long long m)
{
    int score = 0;
    while (m && score < 6) {
        const int hi = 63 - __builtin_clzll(m);
        m &= ~(1ull << hi);
        const int run = m ? hi - (63 - __builtin_clzll(m)) - 1 : hi;
        score += run < 4 ? 3 : run < 12 ? 2 : run < 32 ? 1 : 0;
    }
    return score;
}

// Z layout (lane = 4x4 block row of 4 pixels) -> R8 layout (lane = 8x8 block row of 8 pixels)
__device__ __forceinline__ void z_to_r8(uint32_t z, int lane, uint32_t &lo, uint32_t &hi)
{
    const int i8 = (lane >> 3) & 3, r = lane & 7;
    const int src = i8 * 16 + (r >> 2) * 8 + (r & 3);
    lo = (uint32_t)__shfl((int)z, src); hi = (uint32_t)__shfl((int)z, src + 4);
}

// Half share of the un-normalised SA8D sum of this lane's 8x8 block (R8 layout): the sum over the block's 8 lanes
// times 2 is sum|H8 (e - p) H8|.  Packed-16 butterflies (|coefficient| <= 64*255 fits), last horizontal stage
// replaced by |a+b| + |a-b| = 2 max(|a|,|b|) (see satd4_half_pk).
__device__ __forceinline__ int sa8d_r8_half(uint32_t elo, uint32_t ehi, uint32_t plo, uint32_t phi, int lane)
{
    const s16x2 a = pk_even(elo) - pk_even(plo), b = pk_odd(elo) - pk_odd(plo);       // (d0,d2) (d1,d3)
    const s16x2 c = pk_even(ehi) - pk_even(phi), d = pk_odd(ehi) - pk_odd(phi);       // (d4,d6) (d5,d7)
    const s16x2 s0 = a + b, s1 = a - b, s2 = c + d, s3 = c - d;                       // index bit 0
    s16x2 t[4] = { s0 + s2, s0 - s2, s1 + s3, s1 - s3 };                              // index bit 2
    const s16x2 sg1 = pk_sign(lane & 1), sg2 = pk_sign(lane & 2), sg4 = pk_sign(lane & 4);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s16x2 u = pk_bfly<DPP_XOR1>(t[i], sg1);
        u = pk_bfly<DPP_XOR2>(u, sg2);
        u = u * sg4 + as_s16x2((uint32_t)xor4((int)as_u32(u)));
        u = __builtin_elementwise_max(u, -u);
        const s16x2 m = __builtin_elementwise_max(u, as_s16x2(__builtin_amdgcn_alignbit(as_u32(u), as_u32(u), 16)));   // index bit 1
        acc += as_u32(m) & 0xffffu;
    }
    return (int)acc;
}

// pixel_hadamard_ac ([x264-upstream] common/pixel.c, psy-RD: ssd_plane uses it for every size from 8x8 up): this lane's share of
// sum|H4 x H4| - sum(x) over the 4x4 blocks (e4) and of sum|H8 x H8| - sum(x) over the 8x8 blocks (e8) of a 16x16 block; the caller sums over
// the wave and shifts by 1 / 2.  All 64 lanes must be active (DPP butterflies and the layout change).
__device__ __forceinline__ void psy_energy_r8(uint32_t lo, uint32_t hi, int lane, int &e4, int &e8)      // R8 layout, lanes < 32 carry pixels
{
    const int pix = (int)__builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
    e4 = 2 * (satd4_half(lo, 0u, lane) + satd4_half(hi, 0u, lane)) - pix;
    e8 = 2 * sa8d_r8_half(lo, hi, 0u, 0u, lane) - pix;
}
__device__ __forceinline__ void psy_energy_z(uint32_t z, int lane, int &e4, int &e8)                      // Z layout, 64 lanes x 4 pixels
{
    e4 = 2 * satd4_half(z, 0u, lane) - (int)__builtin_amdgcn_sad_u8(z, 0u, 0u);
    uint32_t lo, hi;
    z_to_r8(z, lane, lo, hi);
    const int h8 = 2 * sa8d_r8_half(lo, hi, 0u, 0u, lane) - (int)__builtin_amdgcn_sad_u8(lo, 0u, __builtin_amdgcn_sad_u8(hi, 0u, 0u));
    e8 = lane < 32 ? h8 : 0;
}

}  // namespace x264gpu

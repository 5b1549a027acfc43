// dsp.hip.h — wave64 device library for the H.264 encode hot path on gfx950 (CDNA4).
//
// Layout convention ("Z layout"): one wavefront owns one 16x16 luma macroblock.  Lane l holds one
// 4-pixel row segment: blk = l>>2 is the 4x4 block in H.264/x264 block order (zigzag of 8x8s, so
// lanes 16k..16k+15 are exactly 8x8 block k == one DPP row), j = l&3 is the row inside the block.
// A 4x4 block is therefore one *quad* of lanes and every 4x4 transform is in-lane butterflies plus
// quad_perm DPP transposes — no LDS, no MFMA (these are int16 add/sub/abs paths, not GEMMs).
//
// Arithmetic restates the oracle (oracle/*.c) bit-exactly; file:line citations live there.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace x264gpu {

// the lane index once more, as a value the compiler cannot trace back to the first one: whatever is derived from it (masks, LDS offsets, Z-layout
// coordinates) is recomputed after this point instead of being hoisted out of the macroblock loop and kept alive — in scratch memory — across it
// Orders this wavefront's LDS accesses: what one lane wrote before is what another lane reads after.  A workgroup here is ONE wavefront and the LDS
// pipeline takes a wavefront's instructions in issue order, so the hardware needs nothing — the compiler must only keep the accesses in program order
// (LDS_ORDER_WAIT: the earlier form, which also drained the LDS queue, i.e. stalled the wavefront for every store's round trip).
__device__ __forceinline__ void lds_order()
{
#ifdef LDS_ORDER_WAIT
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}
__device__ __forceinline__ int relane(int lane0) { int l = lane0; asm volatile("" : "+v"(l)); __builtin_assume(l >= 0 && l < 64); return l; }

// ---------------------------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------------------------
#define X264GPU_QUAD(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
enum : int {
    DPP_XOR1 = X264GPU_QUAD(1, 0, 3, 2),
    DPP_XOR2 = X264GPU_QUAD(2, 3, 0, 1),
    DPP_BC0 = X264GPU_QUAD(0, 0, 0, 0),
    DPP_BC1 = X264GPU_QUAD(1, 1, 1, 1),
    DPP_BC2 = X264GPU_QUAD(2, 2, 2, 2),
    DPP_BC3 = X264GPU_QUAD(3, 3, 3, 3),
    DPP_ROW_MIRROR = 0x140,
    DPP_ROW_HALF_MIRROR = 0x141,
};

template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// sum over the 4 lanes of a quad (result in every lane of the quad)
__device__ __forceinline__ int quad_sum(int v)
{
    v += dpp<DPP_XOR1>(v);
    v += dpp<DPP_XOR2>(v);
    return v;
}
// sum over each DPP row of 16 lanes (result in every lane of the row)
__device__ __forceinline__ int row16_sum(int v)
{
    v = quad_sum(v);
    v += dpp<DPP_ROW_HALF_MIRROR>(v);
    v += dpp<DPP_ROW_MIRROR>(v);
    return v;
}
// sum over the whole wave; result is wave-uniform
__device__ __forceinline__ int wave_sum(int v)
{
    v = row16_sum(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int quad_or(int v)
{
    v |= dpp<DPP_XOR1>(v);
    v |= dpp<DPP_XOR2>(v);
    return v;
}
__device__ __forceinline__ int row16_or(int v)
{
    v = quad_or(v);
    v |= dpp<DPP_ROW_HALF_MIRROR>(v);
    v |= dpp<DPP_ROW_MIRROR>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    unsigned t;
    t = (unsigned)dpp<DPP_XOR1>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp<DPP_XOR2>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp<DPP_ROW_MIRROR>((int)v); v = t < v ? t : v;
    unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}

// the same for a value that is already uniform within every DPP row of 16 lanes (one search candidate per row): four lanes tell it all
__device__ __forceinline__ unsigned rows_min_u32(unsigned v)
{
    unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}
// ... uniform within every eight lanes (eight candidates of eight rows)
__device__ __forceinline__ unsigned halfrows_min_u32(unsigned v)
{
    const unsigned t = (unsigned)dpp<DPP_ROW_MIRROR>((int)v);
    return rows_min_u32(t < v ? t : v);
}

__device__ __forceinline__ unsigned row16_min_u32(unsigned v)      // min over a DPP row of quad-uniform values
{
    unsigned t;
    t = (unsigned)dpp<DPP_ROW_HALF_MIRROR>((int)v); v = t < v ? t : v;
    t = (unsigned)dpp<DPP_ROW_MIRROR>((int)v); v = t < v ? t : v;
    return v;
}


// transpose a 4x4 tile held as 4 registers x 4 quad lanes (lane j, reg c) -> (lane c, reg j)
__device__ __forceinline__ void quad_transpose(int v[4], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
    // exchange 1x1 sub-blocks between lane pairs (j, j^1)
    int s0 = o1 ? v[0] : v[1], s1 = o1 ? v[2] : v[3];
    int r0 = dpp<DPP_XOR1>(s0), r1 = dpp<DPP_XOR1>(s1);
    if (o1) { v[0] = r0; v[2] = r1; } else { v[1] = r0; v[3] = r1; }
    // exchange 2x2 sub-blocks between lane pairs (j, j^2)
    s0 = o2 ? v[0] : v[2]; s1 = o2 ? v[1] : v[3];
    r0 = dpp<DPP_XOR2>(s0); r1 = dpp<DPP_XOR2>(s1);
    if (o2) { v[0] = r0; v[1] = r1; } else { v[2] = r0; v[3] = r1; }
}

// ---------------------------------------------------------------------------------------------
// Z layout geometry
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int z_blk(int lane) { return lane >> 2; }
__device__ __forceinline__ int z_bx(int blk) { return ((blk >> 2) & 1) * 2 + (blk & 1); }
__device__ __forceinline__ int z_by(int blk) { return ((blk >> 3) & 1) * 2 + ((blk >> 1) & 1); }
__device__ __forceinline__ int z_x0(int lane) { return z_bx(lane >> 2) * 4; }
__device__ __forceinline__ int z_y(int lane) { return z_by(lane >> 2) * 4 + (lane & 3); }

__device__ __forceinline__ void unpack4(uint32_t p, int v[4])
{
    v[0] = p & 0xff; v[1] = (p >> 8) & 0xff; v[2] = (p >> 16) & 0xff; v[3] = p >> 24;
}
__device__ __forceinline__ int clip_u8(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
__device__ __forceinline__ uint32_t pack4_clip(const int v[4])
{
    return (uint32_t)clip_u8(v[0]) | ((uint32_t)clip_u8(v[1]) << 8) | ((uint32_t)clip_u8(v[2]) << 16) |
           ((uint32_t)clip_u8(v[3]) << 24);
}
__device__ __forceinline__ uint32_t pack4(const int v[4])
{
    return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
// rounding byte-wise average of 4 packed pixels: (a+b+1)>>1 per byte
__device__ __forceinline__ uint32_t avg4_u8(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_lerp(a, b, 0x01010101u);          // v_lerp_u8: (a + b + (c & 1)) >> 1 per byte
}

// ---------------------------------------------------------------------------------------------
// metrics (oracle/pixel.c)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sad4(uint32_t a, uint32_t b) { return (int)__builtin_amdgcn_sad_u8(a, b, 0u); }
// sum of squared differences of four packed pixels
__device__ __forceinline__ int ssd4_u8(uint32_t a, uint32_t b)
{
    int s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { const int d = (int)((a >> (8 * i)) & 255) - (int)((b >> (8 * i)) & 255); s += d * d; }
    return s;
}

// per-lane share of sum|H4 d H4| for the quad's 4x4 block (sum over the quad, then >>1, is SATD)
__device__ __forceinline__ int satd_quad_partial(const int d[4], int lane)
{
    int s01 = d[0] + d[1], d01 = d[0] - d[1], s23 = d[2] + d[3], d23 = d[2] - d[3];
    int t[4] = { s01 + s23, s01 - s23, d01 - d23, d01 + d23 };
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int y = dpp<DPP_XOR1>(t[i]);
        t[i] = o1 ? y - t[i] : t[i] + y;
        y = dpp<DPP_XOR2>(t[i]);
        t[i] = o2 ? y - t[i] : t[i] + y;
    }
    return abs(t[0]) + abs(t[1]) + abs(t[2]) + abs(t[3]);
}

// ---- packed-16 SATD ------------------------------------------------------------------------------------
// The 4x4 Hadamard of pixel differences never leaves 13 bits (16 * 255), so two coefficients share a VGPR
// and v_pk_* instructions do two butterflies at once.  Two identities remove most of the rest:
//   * any permutation of the four inputs of a 4-point Hadamard only permutes/negates its outputs, so the
//     byte pairs (0,2) / (1,3) that one AND / one shift+AND extract can be used as they come;
//   * |a+b| + |a-b| = 2*max(|a|,|b|), so the last horizontal stage is a max, and because a block's 16
//     coefficients all have the parity of its pixel sum, sum|c| is even: the per-lane "half share" below
//     summed over the quad IS the block's SATD (sum|c| >> 1), exactly.
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ s16x2 pk_even(uint32_t p) { return as_s16x2(p & 0x00ff00ffu); }          // (p0, p2)
__device__ __forceinline__ s16x2 pk_odd(uint32_t p) { return as_s16x2((p >> 8) & 0x00ff00ffu); }    // (p1, p3)
// per-lane vertical butterfly signs: +1 on the lane that adds, -1 on the lane that subtracts
__device__ __forceinline__ s16x2 pk_sign(bool neg) { return as_s16x2(neg ? 0xffffffffu : 0x00010001u); }
template <int CTRL>
__device__ __forceinline__ s16x2 pk_bfly(s16x2 v, s16x2 sg)
{
    const s16x2 y = as_s16x2((uint32_t)dpp<CTRL>((int)as_u32(v)));
    return v * sg + y;                                             // v_pk_mad_i16
}
// half share of sum|H4 (e - p) H4| for one 4x4 block: lane = one row (4 packed pixels), quad = block
// da, db: the row's four pixel differences as two packed pairs, in any pairing (input-permutation invariance)
__device__ __forceinline__ int satd4_half_diff(s16x2 da, s16x2 db, s16x2 sg1, s16x2 sg2)
{
    s16x2 u = da + db, v = da - db;                                // first horizontal stage
    u = pk_bfly<DPP_XOR1>(u, sg1); v = pk_bfly<DPP_XOR1>(v, sg1);  // both vertical stages across the quad
    u = pk_bfly<DPP_XOR2>(u, sg2); v = pk_bfly<DPP_XOR2>(v, sg2);
    u = __builtin_elementwise_max(u, -u); v = __builtin_elementwise_max(v, -v);
    const s16x2 m = __builtin_elementwise_max(u, as_s16x2(__builtin_amdgcn_alignbit(as_u32(u), as_u32(u), 16))) +
                    __builtin_elementwise_max(v, as_s16x2(__builtin_amdgcn_alignbit(as_u32(v), as_u32(v), 16)));
    return (int)(as_u32(m) & 0xffffu);
}
__device__ __forceinline__ int satd4_half_pk(s16x2 e_even, s16x2 e_odd, uint32_t p, s16x2 sg1, s16x2 sg2)
{
    const s16x2 da = e_even - pk_even(p), db = e_odd - pk_odd(p);
    s16x2 u = da + db, v = da - db;                                // first horizontal stage
    u = pk_bfly<DPP_XOR1>(u, sg1); v = pk_bfly<DPP_XOR1>(v, sg1);  // both vertical stages across the quad
    u = pk_bfly<DPP_XOR2>(u, sg2); v = pk_bfly<DPP_XOR2>(v, sg2);
    u = __builtin_elementwise_max(u, -u); v = __builtin_elementwise_max(v, -v);
    const s16x2 m = __builtin_elementwise_max(u, as_s16x2(__builtin_amdgcn_alignbit(as_u32(u), as_u32(u), 16))) +
                    __builtin_elementwise_max(v, as_s16x2(__builtin_amdgcn_alignbit(as_u32(v), as_u32(v), 16)));
    return (int)(as_u32(m) & 0xffffu);
}
__device__ __forceinline__ int satd4_half(uint32_t e, uint32_t p, int lane)
{
    return satd4_half_pk(pk_even(e), pk_odd(e), p, pk_sign(lane & 1), pk_sign(lane & 2));
}
// lane = one 16-pixel row (4 dwords); the quad covers four 4x4 blocks side by side: half share of their SATDs
template <int NB = 4>          // NB dwords of the row carry pixels (2: an 8-pixel row)
__device__ __forceinline__ int satd16x4_half_pk(const uint32_t e[4], const uint32_t p[4], s16x2 sg1, s16x2 sg2)
{
    s16x2 acc = as_s16x2(0u);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const s16x2 da = pk_even(e[b]) - pk_even(p[b]), db = pk_odd(e[b]) - pk_odd(p[b]);
        s16x2 u = da + db, v = da - db;
        u = pk_bfly<DPP_XOR1>(u, sg1); v = pk_bfly<DPP_XOR1>(v, sg1);
        u = pk_bfly<DPP_XOR2>(u, sg2); v = pk_bfly<DPP_XOR2>(v, sg2);
        u = __builtin_elementwise_max(u, -u); v = __builtin_elementwise_max(v, -v);
        acc += __builtin_elementwise_max(u, as_s16x2(__builtin_amdgcn_alignbit(as_u32(u), as_u32(u), 16))) +
               __builtin_elementwise_max(v, as_s16x2(__builtin_amdgcn_alignbit(as_u32(v), as_u32(v), 16)));   // <= 8 * 4080
    }
    return (int)(as_u32(acc) & 0xffffu);
}

// 2-D 4x4 Hadamard kept in registers (natural layout in, same layout out, unnormalised)
__device__ __forceinline__ void hadamard4_quad(int t[4], int lane)
{
    int s01 = t[0] + t[1], d01 = t[0] - t[1], s23 = t[2] + t[3], d23 = t[2] - t[3];
    t[0] = s01 + s23; t[1] = s01 - s23; t[2] = d01 - d23; t[3] = d01 + d23;
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int y = dpp<DPP_XOR1>(t[i]);
        t[i] = o1 ? y - t[i] : t[i] + y;
        y = dpp<DPP_XOR2>(t[i]);
        t[i] = o2 ? y - t[i] : t[i] + y;
    }
}

// ---------------------------------------------------------------------------------------------
// transforms (oracle/dct.c).  "natural layout": quad lane r = coefficient/pixel row, reg c = column
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void fwd4_1d(int v[4])
{
    int s03 = v[0] + v[3], s12 = v[1] + v[2], d03 = v[0] - v[3], d12 = v[1] - v[2];
    v[0] = s03 + s12; v[1] = 2 * d03 + d12; v[2] = s03 - s12; v[3] = d03 - 2 * d12;
}
__device__ __forceinline__ void inv4_1d(int v[4])
{
    int e0 = v[0] + v[2], e1 = v[0] - v[2], e2 = (v[1] >> 1) - v[3], e3 = v[1] + (v[3] >> 1);
    v[0] = e0 + e3; v[1] = e1 + e2; v[2] = e1 - e2; v[3] = e0 - e3;
}
__device__ __forceinline__ void had4_1d(int v[4])
{
    int s01 = v[0] + v[1], d01 = v[0] - v[1], s23 = v[2] + v[3], d23 = v[2] - v[3];
    v[0] = s01 + s23; v[1] = s01 - s23; v[2] = d01 - d23; v[3] = d01 + d23;
}
// residual rows -> coefficients (sub4x4_dct)
__device__ __forceinline__ void dct4_quad(int v[4], int lane)
{
    fwd4_1d(v);
    quad_transpose(v, lane);
    fwd4_1d(v);
    quad_transpose(v, lane);
}
// coefficients -> residual rows, already (x+32)>>6 (8.5.12: rows first, then columns)
__device__ __forceinline__ void idct4_quad(int v[4], int lane)
{
    inv4_1d(v);
    quad_transpose(v, lane);
    inv4_1d(v);
    quad_transpose(v, lane);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (v[i] + 32) >> 6;
}
__device__ __forceinline__ void had4x4_quad(int v[4], int lane)
{
    had4_1d(v);
    quad_transpose(v, lane);
    had4_1d(v);
    quad_transpose(v, lane);
}

// ---------------------------------------------------------------------------------------------
// quant / dequant with flat CQM: a 4x4 table has only three distinct values, selected by the
// parity class of (row, col): 0 = (even,even), 1 = mixed, 2 = (odd,odd)   (oracle/quant.c)
// ---------------------------------------------------------------------------------------------
struct Q4 {          // everything a 4x4 block needs for one (qp, list)
    int mf[3];       // quant multipliers
    int bias[3];     // deadzone biases
    int dq[3];       // LevelScale4x4 = 16*normAdjust
    int qp;
};

__device__ __forceinline__ int quant_one(int c, int mf, int bias)
{
    int a = abs(c);
    int l = ((bias + a) * mf) >> 16;
    return c > 0 ? l : -l;
}
__device__ __forceinline__ int dequant_one(int l, int dq, int qbits)
{
    return qbits >= 0 ? (l * dq) << qbits : (l * dq + (1 << (-qbits - 1))) >> -qbits;
}
// returns nonzero flag for this lane's row
__device__ __forceinline__ int quant4_row(int v[4], const Q4 &q, int row)
{
    const bool odd = row & 1;
    int mfe = odd ? q.mf[1] : q.mf[0], mfo = odd ? q.mf[2] : q.mf[1];
    int be = odd ? q.bias[1] : q.bias[0], bo = odd ? q.bias[2] : q.bias[1];
    v[0] = quant_one(v[0], mfe, be); v[1] = quant_one(v[1], mfo, bo);
    v[2] = quant_one(v[2], mfe, be); v[3] = quant_one(v[3], mfo, bo);
    return (v[0] | v[1] | v[2] | v[3]) != 0;
}
__device__ __forceinline__ void dequant4_row(int v[4], const Q4 &q, int row)
{
    const bool odd = row & 1;
    int de = odd ? q.dq[1] : q.dq[0], dod = odd ? q.dq[2] : q.dq[1];
    int qb = q.qp / 6 - 4;
    v[0] = dequant_one(v[0], de, qb); v[1] = dequant_one(v[1], dod, qb);
    v[2] = dequant_one(v[2], de, qb); v[3] = dequant_one(v[3], dod, qb);
}

// scan index (zigzag) of raster position r*4+c
__device__ __forceinline__ int zigzag4_inv(int pos)
{
    // inverse of {0,1,4,8,5,2,3,6,9,12,13,10,7,11,14,15}, packed 4 bits per entry
    const uint64_t tab = 0xFEA9DB83C7426510ull;   // tab[pos] = scan index
    return (int)((tab >> (pos * 4)) & 15);
}

// dct-decimate score of a 4x4 block from its scan-order nonzero mask (all |level| <= 1 assumed);
// `first` = lowest scan index that belongs to the block (1 for AC-only blocks)
__device__ __forceinline__ int decimate_from_mask(unsigned mask, int first)
{
    int score = 0;
    while (mask) {
        int p = 31 - __builtin_clz(mask);
        mask &= ~(1u << p);
        int nxt = mask ? 31 - __builtin_clz(mask) : first - 1;
        int run = p - nxt - 1;
        score += run < 1 ? 3 : run < 3 ? 2 : run < 6 ? 1 : 0;
    }
    return score;
}

}  // namespace x264gpu

// k_analyse.hip.h — motion-estimation building blocks (A2/A3): LDS search window SADs, the staged sub-pel neighbourhood, the
// chroma-ME term, and the candidate-parallel partition search the lookahead kernel uses.  The macroblock loop of the frame
// pipeline is k_mb.hip.h.
//
//   full-pel:  the reference search window (50 rows x 64 B, centred on the best start predictor) is
//              staged in LDS with coalesced 8-byte loads; candidates are evaluated four at a time,
//              lane = (candidate, row): 16 pixels per lane via v_alignbyte + v_sad_u8, 16-lane DPP row
//              reduction, then a packed (cost<<3|tag) wave min replaying x264's first-best tie-breaks.
//   sub-pel:   half-pel diamond on SAD, quarter-pel diamond on SATD; the +-2 px neighbourhood of the full-pel
//              winner in all four half-pel planes is staged in LDS once (sub_stage), SATD in the Z layout.
#pragma once
#include "enc_common.hip.h"

namespace x264gpu {

// occupancy the register allocator targets for k_analyse_p (waves per SIMD)
#ifndef X264GPU_ANALYSE_WAVES
#define X264GPU_ANALYSE_WAVES 4
#endif
#ifndef X264GPU_ANALYSE_WAVES_MIXED
#define X264GPU_ANALYSE_WAVES_MIXED 3      // the mixed-refs instantiation carries more per-lane state: 3 waves/SIMD (170 VGPRs) beats spilling at 4
#endif

constexpr int WIN_ROWS = 50, WIN_COLS = 64, WIN_STRIDE = 68, WIN_R = 17;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }
__device__ __forceinline__ int median3(int a, int b, int c)
{
    int mn = min(a, b), mx = max(a, b);
    return c < mn ? mn : c > mx ? mx : c;
}

// SAD of one 16-pixel row held in cr[4] against 16 bytes at (row base `wrow`, byte offset xoff) in LDS
__device__ __forceinline__ int sad_row16_lds(const uint8_t *wrow, int xoff, const uint32_t cr[4])
{
    const uint32_t *w = (const uint32_t *)(wrow + (xoff & ~3));
    const int sh = xoff & 3;
    uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    unsigned s = 0;
    s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1, w0, sh), cr[0], s);
    s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w2, w1, sh), cr[1], s);
    s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w3, w2, sh), cr[2], s);
    s = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w4, w3, sh), cr[3], s);
    return (int)s;
}
// the 16 bytes at (row base `wrow`, byte offset xoff) in LDS
__device__ __forceinline__ void row16_lds(const uint8_t *wrow, int xoff, uint32_t out[4])
{
    const uint32_t *w = (const uint32_t *)(wrow + (xoff & ~3));
    const int sh = xoff & 3;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    out[0] = __builtin_amdgcn_alignbyte(w1, w0, sh); out[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
    out[2] = __builtin_amdgcn_alignbyte(w3, w2, sh); out[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
}
__device__ __forceinline__ int sad_row16_global(const uint8_t *p, const uint32_t cr[4])
{
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s = __builtin_amdgcn_sad_u8(load_u32_unaligned(p + 4 * i), cr[i], s);
    return (int)s;
}

__device__ __forceinline__ int hex_dx(int i) { return (int)((0x01343101u >> (4 * i)) & 15) - 2; }
__device__ __forceinline__ int hex_dy(int i) { return (int)((0x20024420u >> (4 * i)) & 15) - 2; }
__device__ __forceinline__ int sq_dx(int k) { return (int)((0x22002011u >> (4 * (k - 1))) & 15) - 1; }  // k = 1..8
__device__ __forceinline__ int sq_dy(int k) { return (int)((0x20201120u >> (4 * (k - 1))) & 15) - 1; }

// ------------------------------------------------------------------------------------------------
// Sub-pel neighbourhood in LDS.  Once a full-pel vector is fixed, every sample the half-/quarter-pel
// diamonds can touch lies within +-2 px of the displaced block, in one of the four half-pel planes.  The
// lanes of a partition copy that neighbourhood ((h+4) rows x (w+4 rounded to dwords) of all four planes) to
// LDS with one batch of independent aligned dword loads; the refinement then runs without touching HBM/L2.
//   geometry: rows `rh`, row pitch 1<<rwl dwords, `ncol` dwords per row actually loaded, n = rh<<rwl dwords
//   per plane; origin (x0,y0) in picture coordinates, x0 a multiple of 4.
// ------------------------------------------------------------------------------------------------
// M = margin in pixels: 2 covers subme <= 7 (2 half-pel + 3 quarter-pel steps), 5 covers the 4 + 10 steps of subme >= 8.
template <int M> struct SubGeo {
    static constexpr int DWORDS = M == 2 ? 768 : 2304;            // 4 parts x 4 planes x (8+2M) rows x pitch (8x8 is the largest)
    static __device__ __forceinline__ int rwl(int w) { return M == 2 && w == 8 ? 2 : 3; }
    static __device__ __forceinline__ int ncol(int w) { return (w + 2 * M + 6) >> 2; }
    static __device__ __forceinline__ int rh(int h) { return h + 2 * M; }
};
template <int M>
__device__ __forceinline__ void sub_stage(uint32_t *buf, const uint8_t *__restrict__ p00, size_t pb, int rs, int x0, int y0,
                                          int rwl, int rh, int ncol, int li, int L)
{
    const int n = rh << rwl;
    if (M == 2) {                                                    // n <= 3*L for every shape: 12 loads in flight, then 12 stores
        uint32_t v[12];
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = li + t * L, row = i >> rwl, col = i & ((1 << rwl) - 1);
                v[pl * 3 + t] = (i < n && col < ncol) ? *(const uint32_t *)(p00 + pl * pb + (long)(y0 + row) * rs + x0 + 4 * col) : 0u;
            }
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int t = 0; t < 3; t++) {
                const int i = li + t * L;
                if (i < n) buf[pl * n + i] = v[pl * 3 + t];
            }
    } else {
        for (int i = li; i < n; i += L) {
            const int row = i >> rwl, col = i & ((1 << rwl) - 1);
            uint32_t v[4];
#pragma unroll
            for (int pl = 0; pl < 4; pl++) v[pl] = col < ncol ? *(const uint32_t *)(p00 + pl * pb + (long)(y0 + row) * rs + x0 + 4 * col) : 0u;
#pragma unroll
            for (int pl = 0; pl < 4; pl++) buf[pl * n + i] = v[pl];
        }
    }
}
// sub_stage<2> in two halves: the twelve requests, and (after other work has covered their latency) the copy into LDS
__device__ __forceinline__ void sub_issue2(uint32_t v[12], const uint8_t *__restrict__ p00, size_t pb, int rs, int x0, int y0, int rwl, int rh, int ncol, int li)
{
    const int n = rh << rwl;
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int i = li + t * 64, row = i >> rwl, col = i & ((1 << rwl) - 1);
            v[pl * 3 + t] = (i < n && col < ncol) ? *(const uint32_t *)(p00 + pl * pb + (long)(y0 + row) * rs + x0 + 4 * col) : 0u;
        }
}
__device__ __forceinline__ void sub_commit2(uint32_t *buf, const uint32_t v[12], int rwl, int rh, int li)
{
    const int n = rh << rwl;
#pragma unroll
    for (int pl = 0; pl < 4; pl++)
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int i = li + t * 64;
            if (i < n) buf[pl * n + i] = v[pl * 3 + t];
        }
}
__device__ __forceinline__ uint32_t lds_u32_at(const uint32_t *buf, int byte_off)
{
    const uint32_t *w = buf + (byte_off >> 2);
    return __builtin_amdgcn_alignbyte(w[1], w[0], byte_off & 3);
}
// mc_luma_row4 on the staged neighbourhood: 4 pixels at picture position (x..x+3, y) displaced by (mvx,mvy) qpel
__device__ __forceinline__ uint32_t sub_row4(const uint32_t *buf, int n, int rwl, int x0, int y0, int x, int y, int mvx, int mvy)
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    const int bx = x + (mvx >> 2) - x0, by = y + (mvy >> 2) - y0;
    const uint32_t a = lds_u32_at(buf + pl0 * n, (((by + ((mvy & 3) == 3 ? 1 : 0)) << rwl) << 2) + bx);
    const uint32_t b = lds_u32_at(buf + pl1 * n, ((by << rwl) << 2) + bx + ((mvx & 3) == 3 ? 1 : 0));
    return (idx & 5) ? avg4_u8(a, b) : a;
}
// 16 pixels of row y at picture x..x+15 displaced by the quarter-pel vector (mvx,mvy): out[4] packed dwords
__device__ __forceinline__ void sub_row16(const uint32_t *buf, int n, int rwl, int x0, int y0, int x, int y, int mvx, int mvy, uint32_t out[4])
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    const int bx = x + (mvx >> 2) - x0, by = y + (mvy >> 2) - y0;
    const int o0 = (((by + ((mvy & 3) == 3 ? 1 : 0)) << rwl) << 2) + bx, o1 = ((by << rwl) << 2) + bx + ((mvx & 3) == 3 ? 1 : 0);
    const uint32_t *wa = buf + pl0 * n + (o0 >> 2), *wb = buf + pl1 * n + (o1 >> 2);
    uint32_t a[5], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { a[i] = wa[i]; b[i] = wb[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t pa = __builtin_amdgcn_alignbyte(a[i + 1], a[i], o0 & 3), pb_ = __builtin_amdgcn_alignbyte(b[i + 1], b[i], o1 & 3);
        out[i] = (idx & 5) ? avg4_u8(pa, pb_) : pa;
    }
}
// SAD of a 16-pixel row at a HALF-pel displacement (single plane, no averaging) against cr[4]
__device__ __forceinline__ int sub_sad_row16_hpel(const uint32_t *buf, int n, int rwl, int x0, int y0, int x, int y, int mvx, int mvy,
                                                  const uint32_t cr[4])
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3;
    const int off = (((y + (mvy >> 2) - y0) << rwl) << 2) + x + (mvx >> 2) - x0;
    const uint32_t *w = buf + pl0 * n + (off >> 2);
    const int sh = off & 3;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
    unsigned sd = 0;
    sd = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1, w0, sh), cr[0], sd);
    sd = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w2, w1, sh), cr[1], sd);
    sd = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w3, w2, sh), cr[2], sd);
    sd = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w4, w3, sh), cr[3], sd);
    return (int)sd;
}

// ------------------------------------------------------------------------------------------------
// Chroma term of the sub-pel SATD costs (oracle chroma_me_satd; x264 COST_MV_SATD under b_chroma_me).  A lane owns one row of
// a 4x4 chroma block of BOTH planes (the NV12 bytes it loads serve U and V), a quad of lanes the block.  Bilinear 1/8-pel
// prediction and Hadamard run on packed pairs: two neighbouring samples of a plane share a VGPR (weights <= 64, sums < 2^15).
// Returns the lane's half share of SATD(U) + SATD(V); the sum over a candidate's lanes is the chroma cost of the candidate.
// (cx, cy): chroma-sample position of the lane's four pixels; e0/e1: their 8 source bytes U0 V0 U1 V1 | U2 V2 U3 V3.
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct ChromaTaps { uint32_t a0, a1, a2, b0, b1, b2; };         // the six NV12 dwords one lane needs for one candidate
// explicit weighted prediction of two chroma samples held as 16-bit lanes (wpk = EncK::wc0 packing)
__device__ __forceinline__ u16x2 wp2(u16x2 p, int wpk)
{
    const int offset = (int)(int8_t)(wpk & 0xff), scale = (int)(int8_t)((wpk >> 8) & 0xff), denom = (wpk >> 16) & 0xff;
    const int rnd = denom ? 1 << (denom - 1) : 0;
    int a = (((int)p.x * scale + rnd) >> denom) + offset, b = (((int)p.y * scale + rnd) >> denom) + offset;
    a = a < 0 ? 0 : a > 255 ? 255 : a; b = b < 0 ? 0 : b > 255 ? 255 : b;
    u16x2 r; r.x = (unsigned short)a; r.y = (unsigned short)b;
    return r;
}
// wcu / wcv: the reference's explicit Cb / Cr weight (0 = none; wave-uniform)
template <bool WT = false>
__device__ __forceinline__ int chroma_me_cost(const ChromaTaps &t, int mvx, int mvy, uint32_t e0, uint32_t e1, s16x2 sg1, s16x2 sg2, int wcu = 0, int wcv = 0)
{
    const int dx = mvx & 7, dy = mvy & 7;
    const uint32_t wA = (uint32_t)((8 - dx) * (8 - dy)) * 0x10001u, wB = (uint32_t)(dx * (8 - dy)) * 0x10001u;
    const uint32_t wC = (uint32_t)((8 - dx) * dy) * 0x10001u, wD = (uint32_t)(dx * dy) * 0x10001u;
    const uint32_t a0 = t.a0, a1 = t.a1, a2 = t.a2, b0 = t.b0, b1 = t.b1, b2 = t.b2;
    const uint32_t a01 = __builtin_amdgcn_alignbyte(a1, a0, 2), a12 = __builtin_amdgcn_alignbyte(a2, a1, 2);     // one sample to the right
    const uint32_t b01 = __builtin_amdgcn_alignbyte(b1, b0, 2), b12 = __builtin_amdgcn_alignbyte(b2, b1, 2);
    int acc = 0;
#pragma unroll
    for (int sh = 0; sh < 16; sh += 8) {
#define CH(x) __builtin_bit_cast(u16x2, ((x) >> sh) & 0x00ff00ffu)
#define W(x) __builtin_bit_cast(u16x2, x)
        u16x2 p01 = (W(wA) * CH(a0) + W(wB) * CH(a01) + W(wC) * CH(b0) + W(wD) * CH(b01) + W(0x00200020u)) >> 6;
        u16x2 p23 = (W(wA) * CH(a1) + W(wB) * CH(a12) + W(wC) * CH(b1) + W(wD) * CH(b12) + W(0x00200020u)) >> 6;
        if constexpr (WT) { const int wk = sh ? wcv : wcu; if (wk) { p01 = wp2(p01, wk); p23 = wp2(p23, wk); } }
        const s16x2 da = __builtin_bit_cast(s16x2, CH(e0)) - __builtin_bit_cast(s16x2, p01);
        const s16x2 db = __builtin_bit_cast(s16x2, CH(e1)) - __builtin_bit_cast(s16x2, p23);
#undef CH
#undef W
        acc += satd4_half_diff(da, db, sg1, sg2);
    }
    return acc;
}
__device__ __forceinline__ int chroma_me_half(const uint8_t *__restrict__ nv12, int rs, int cx, int cy, int mvx, int mvy, uint32_t e0, uint32_t e1,
                                              s16x2 sg1, s16x2 sg2)
{
    const uint8_t *s = nv12 + (long)(cy + (mvy >> 3)) * rs + 2 * (cx + (mvx >> 3));
    ChromaTaps t;
    t.a0 = load_u32_unaligned(s); t.a1 = load_u32_unaligned(s + 4); t.a2 = load_u32_unaligned(s + 8);
    t.b0 = load_u32_unaligned(s + rs); t.b1 = load_u32_unaligned(s + rs + 4); t.b2 = load_u32_unaligned(s + rs + 8);
    return chroma_me_cost(t, mvx, mvy, e0, e1, sg1, sg2);
}

// ------------------------------------------------------------------------------------------------
// Chroma neighbourhood in LDS: like sub_stage for luma, the NV12 samples every sub-pel candidate of one block / partition can
// touch are staged once after the full-pel search (one global-memory latency instead of one per candidate).  A candidate's
// eighth-pel vector is 4 * full-pel winner + d with |d| <= 6 (M == 2) or 18 (M == 5) quarter-pels, so its integer chroma offset
// lies in [(b >> 1) - MG, (b >> 1) + MG], MG = 1 or 3.  Rows hold ndw dwords (= 2 * ndw chroma samples of both planes),
// starting at an even chroma sample x0c so that the global loads are aligned dwords.
// ------------------------------------------------------------------------------------------------
template <int M> struct CSubGeo {
    static constexpr int MG = M == 2 ? 1 : 3;
    static __device__ __forceinline__ int ndw(int wc) { return (wc + 2 * MG + 2 + 1) >> 1; }    // samples: wc + 2 MG + 1 tap + 1 alignment, two per dword
    static __device__ __forceinline__ int rows(int hc) { return hc + 2 * MG + 1; }
    static constexpr int DWORDS = M == 2 ? 128 : 288;              // four 8x8 partitions are the largest set: 4 x ndw(4) x rows(4), + slack
};
__device__ __forceinline__ void chroma_stage(uint32_t *cb, const uint8_t *__restrict__ nv12, int rs, int x0c, int y0c, int ndw, int nrows, int l, int L)
{
    for (int i = l; i < ndw * nrows; i += L) {
        const int row = i / ndw, col = i - row * ndw;
        cb[i] = *(const uint32_t *)(nv12 + (long)(y0c + row) * rs + 2 * x0c + 4 * col);
    }
}
// chroma_stage in two halves for M == 2 (at most 66 dwords: two per lane)
__device__ __forceinline__ void chroma_issue2(uint32_t v[2], const uint8_t *__restrict__ nv12, int rs, int x0c, int y0c, int ndw, int nrows, int l)
{
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = l + 64 * t, row = i / ndw, col = i - row * ndw;
        v[t] = i < ndw * nrows ? *(const uint32_t *)(nv12 + (long)(y0c + row) * rs + 2 * x0c + 4 * col) : 0u;
    }
}
__device__ __forceinline__ void chroma_commit2(uint32_t *cb, const uint32_t v[2], int ndw, int nrows, int l)
{
#pragma unroll
    for (int t = 0; t < 2; t++) { const int i = l + 64 * t; if (i < ndw * nrows) cb[i] = v[t]; }
}
// chroma_me_half on the staged neighbourhood: (cx, cy) chroma position of the lane's four pixels
template <bool WT = false>
__device__ __forceinline__ int chroma_me_lds(const uint32_t *cb, int ndw, int x0c, int y0c, int cx, int cy, int mvx, int mvy, uint32_t e0, uint32_t e1,
                                             s16x2 sg1, s16x2 sg2, int wcu = 0, int wcv = 0)
{
    const int o = ((cy + (mvy >> 3) - y0c) * ndw << 2) + 2 * (cx + (mvx >> 3) - x0c), sh = o & 3;
    const uint32_t *w = cb + (o >> 2), *v = w + ndw;
    ChromaTaps t;
    { const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3];
      t.a0 = __builtin_amdgcn_alignbyte(d1, d0, sh); t.a1 = __builtin_amdgcn_alignbyte(d2, d1, sh); t.a2 = __builtin_amdgcn_alignbyte(d3, d2, sh); }
    { const uint32_t d0 = v[0], d1 = v[1], d2 = v[2], d3 = v[3];
      t.b0 = __builtin_amdgcn_alignbyte(d1, d0, sh); t.b1 = __builtin_amdgcn_alignbyte(d2, d1, sh); t.b2 = __builtin_amdgcn_alignbyte(d3, d2, sh); }
    return chroma_me_cost<WT>(t, mvx, mvy, e0, e1, sg1, sg2, wcu, wcv);
}

// ------------------------------------------------------------------------------------------------
// Sub-partition search (P16x8 / P8x16 / P8x8): every partition of a shape is searched at the same time, each on its own
// lanes, with partition-private motion state and predicated updates, so the wave runs ONE instruction stream for 2 or 4
// independent hexagon / square / half-pel / quarter-pel searches.  Restates oracle me_search_block for the partitions of a shape.
// ------------------------------------------------------------------------------------------------
struct PartCtx {
    const uint8_t *win; int wx0, wy0;          // LDS search window (full-pel plane) and its picture origin
    const uint16_t *cx, *cy;                   // LDS slices of the mv-cost table, index = qpel mv - cbase + 96
    int cbx, cby;                              // qpel mv at slice centre
    const uint8_t *p00; size_t pb; int rs;     // half-pel planes
    int px, py, zx, zy; uint32_t cz;           // macroblock position, lane position, lane's 4 source pixels
    int fmin0, fmax0, fmin1, fmax1, smin0, smax0, smin1, smax1;
    int me_range, me_method, hp_it, qp_it, lane;
    uint32_t *sub;                             // LDS sub-pel neighbourhood buffer (SUB_DWORDS)
    const uint8_t *fenc; int fs;               // source macroblock (for the candidate-parallel 8x8 search)
    const uint8_t *cref, *fuv; int chroma_me;  // chroma-ME: NV12 reference plane (origin), source NV12 of this macroblock
    uint32_t *csub;                            // chroma-ME: LDS chroma neighbourhood buffer (CSubGeo<M>::DWORDS)
    const uint16_t *gcx, *gcy; int mvp0, mvp1; // UMH: mv-cost table in global memory (index = qpel mv), cost predictor
    int col_x, col_y; bool has_col, la_mode;   // lookahead: co-located start candidate (quarter-pel); la_mode selects its search
};
template <bool UMH>
__device__ __forceinline__ int pc_mvcost(const PartCtx &c, int qx, int qy)
{
    if (UMH) return c.gcx[qx] + c.gcy[qy];                         // UMH ends anywhere: no LDS slice around the start
    return c.cx[qx - c.cbx + 96] + c.cy[qy - c.cby + 96];
}
// ------------------------------------------------------------------------------------------------
// Candidate-parallel partition search: the lanes of a partition split into four groups that evaluate four CANDIDATE vectors at
// once; a lane holds 16 pixels of its partition per candidate (as in the 16x16 search).  Per-partition search state is uniform
// over the partition's lanes.  Same arithmetic and tie-breaks as oracle me_search_block.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sad8_lds(const uint8_t *base, int off, uint32_t e0, uint32_t e1)
{
    const uint32_t *w = (const uint32_t *)(base + (off & ~3));
    const int sh = off & 3;
    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
    unsigned sd = __builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w1, w0, sh), e0, 0u);
    return (int)__builtin_amdgcn_sad_u8(__builtin_amdgcn_alignbyte(w2, w1, sh), e1, sd);
}
// 8 pixels of row y at picture x..x+7 displaced by the quarter-pel vector, from the staged sub-pel neighbourhood
__device__ __forceinline__ void sub_row8(const uint32_t *buf, int n, int rwl, int x0, int y0, int x, int y, int mvx, int mvy, uint32_t out[2])
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    const int bx = x + (mvx >> 2) - x0, by = y + (mvy >> 2) - y0;
    const int o0 = (((by + ((mvy & 3) == 3 ? 1 : 0)) << rwl) << 2) + bx, o1 = ((by << rwl) << 2) + bx + ((mvx & 3) == 3 ? 1 : 0);
    const uint32_t *wa = buf + pl0 * n + (o0 >> 2), *wb = buf + pl1 * n + (o1 >> 2);
    const uint32_t a0 = wa[0], a1 = wa[1], a2 = wa[2], b0 = wb[0], b1 = wb[1], b2 = wb[2];
    const uint32_t pa0 = __builtin_amdgcn_alignbyte(a1, a0, o0 & 3), pa1 = __builtin_amdgcn_alignbyte(a2, a1, o0 & 3);
    const uint32_t pb0 = __builtin_amdgcn_alignbyte(b1, b0, o1 & 3), pb1 = __builtin_amdgcn_alignbyte(b2, b1, o1 & 3);
    out[0] = (idx & 5) ? avg4_u8(pa0, pb0) : pa0;
    out[1] = (idx & 5) ? avg4_u8(pa1, pb1) : pa1;
}
// half share of the SATD of an 8x8 partition held as two rows x 8 pixels per lane (quad = the partition): 4x4 blocks are
// (column half) x (lane pair); vertical butterflies: the two rows in the lane, then lane^1
__device__ __forceinline__ int satd8x8_2rows_half(const uint32_t e0[2], const uint32_t e1[2], const uint32_t p0[2], const uint32_t p1[2], s16x2 sg1)
{
    s16x2 acc = as_s16x2(0u);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const s16x2 da0 = pk_even(e0[b]) - pk_even(p0[b]), db0 = pk_odd(e0[b]) - pk_odd(p0[b]);
        const s16x2 da1 = pk_even(e1[b]) - pk_even(p1[b]), db1 = pk_odd(e1[b]) - pk_odd(p1[b]);
        const s16x2 u0 = da0 + db0, v0 = da0 - db0, u1 = da1 + db1, v1 = da1 - db1;
        s16x2 t[4] = { u0 + u1, u0 - u1, v0 + v1, v0 - v1 };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            s16x2 x = pk_bfly<DPP_XOR1>(t[i], sg1);
            x = __builtin_elementwise_max(x, -x);
            acc += __builtin_elementwise_max(x, as_s16x2(__builtin_amdgcn_alignbit(as_u32(x), as_u32(x), 16)));     // <= 8 * 4080
        }
    }
    return (int)(as_u32(acc) & 0xffffu);
}

// ------------------------------------------------------------------------------------------------
// X264_ME_UMH, full-pel part (oracle me_search_block, me_method 2; [x264-upstream] me.c "Uneven-cross Multi-Hexagon-grid").
// The search roams up to ~70 pixels from its start, so it reads the reference plane and the mv-cost table from global memory
// (L2-resident rows) instead of the LDS window of the hexagon search.  SHAPE 0 is the 16x16 block (16 lanes per candidate, a
// lane = one row as two 8-pixel segments), 1..3 the sub-partition layouts of search_parts; all partitions of a shape run at
// once with partition-private state and predicated updates.  Every update is x264's in-order "strictly better wins": four
// candidates are costed side by side and the minimum of (cost << 2 | order) is compared with the running best.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int sad8_global(const uint8_t *p, uint32_t e0, uint32_t e1)
{
    const unsigned sd = __builtin_amdgcn_sad_u8(load_u32_unaligned(p), e0, 0u);
    return (int)__builtin_amdgcn_sad_u8(load_u32_unaligned(p + 4), e1, sd);
}
__device__ __forceinline__ int umh_hex4_dx(int j) { return (int)((0x6280808080806244ull >> (4 * j)) & 15) - 4; }
__device__ __forceinline__ int umh_hex4_dy(int j) { return (int)((0x7766554433221180ull >> (4 * j)) & 15) - 4; }

// MODE 0: UMH from the start (bx, by, bcost).  MODE 1: plain search on global memory — start candidates (predictor, zero,
// co-located: c.col_*) in order, then hexagon + square with c.me_range (the lookahead's x264_me_search, oracle/lookahead.c la_search).
__device__ __forceinline__ void umh_fullpel(const PartCtx &c, const int SHAPE, const int MODE, const int mvd, int &bx, int &by, int &bcost)
{
    const int lane = relane(c.lane), GL = SHAPE == 0 ? 16 : SHAPE == 3 ? 4 : 8;
    const int part = SHAPE == 0 ? 0 : SHAPE == 3 ? lane >> 4 : lane >> 5, cnd = SHAPE == 0 ? lane >> 4 : SHAPE == 3 ? (lane >> 2) & 3 : (lane >> 3) & 3;
    const int sr = lane & (GL - 1), pbase = lane - cnd * GL - sr;
    const int ox = SHAPE == 3 ? (part & 1) * 8 : SHAPE == 2 ? part * 8 : 0, oy = SHAPE == 3 ? (part >> 1) * 8 : SHAPE == 1 ? part * 8 : 0;
    const bool wide = SHAPE < 2;                     // 16-pixel rows: the two segments are the halves of one row
    const int y0 = oy + (wide ? sr : 2 * sr), y1 = wide ? y0 : y0 + 1, x1 = wide ? 8 : 0;
    uint32_t e00, e01, e10, e11;
    { const uint2 a = *(const uint2 *)(c.fenc + (size_t)y0 * c.fs + ox), b = *(const uint2 *)(c.fenc + (size_t)y1 * c.fs + ox + x1);
      e00 = a.x; e01 = a.y; e10 = b.x; e11 = b.y; }
    const uint8_t *g0 = c.p00 + (long)(c.py + y0) * c.rs + c.px + ox, *g1 = c.p00 + (long)(c.py + y1) * c.rs + c.px + ox + x1;
    auto gsum = [&](int v) { v = quad_sum(v); if (GL >= 8) v += xor4(v); if (GL == 16) v += dpp<DPP_ROW_MIRROR>(v); return v; };
    auto cmin = [&](unsigned k) {
        if (SHAPE == 0) return wave_min_u32(k);
        if (SHAPE == 3) return row16_min_u32(k);
        unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x128, 0xf, 0xf, true);
        k = t < k ? t : k;
        t = (unsigned)__shfl_xor((int)k, 16);
        return t < k ? t : k;
    };
    auto inrange = [&](int mx, int my) { return mx >= c.fmin0 && mx <= c.fmax0 && my >= c.fmin1 && my <= c.fmax1; };
    // cost this lane group's candidate (mx,my) if `valid`; the partition takes the best of its four groups when it beats bcost.
    // Returns the winning group (0..3) or -1.
    auto step = [&](int mx, int my, bool valid) {
        mx = valid ? mx : bx; my = valid ? my : by;                 // keep the loads of masked candidates inside the padded plane
        const long o = (long)my * c.rs + mx;
        const int cst = gsum(sad8_global(g0 + o, e00, e01) + sad8_global(g1 + o, e10, e11)) + c.gcx[mx * 4] + c.gcy[my * 4];
        unsigned kk = valid ? ((unsigned)cst << 2) | (unsigned)cnd : 0xffffffffu;
        kk = cmin(kk);
        const int wl = pbase + (int)(kk & 3) * GL;
        const int wx = __shfl(mx, wl), wy = __shfl(my, wl);
        if (kk != 0xffffffffu && (int)(kk >> 2) < bcost) { bcost = (int)(kk >> 2); bx = wx; by = wy; return (int)(kk & 3); }
        return -1;
    };
    const int d1x = cnd == 2 ? -1 : cnd == 3 ? 1 : 0, d1y = cnd == 0 ? -1 : cnd == 1 ? 1 : 0;      // DIA1: (0,-1) (0,1) (-1,0) (1,0)
    const int shift = SHAPE == 0 ? 0 : SHAPE == 3 ? 2 : 1;                                         // x264 pixel_size_shift
#define UMH_TH(v) (bcost < ((v) >> shift))
    const int pmx = clampi((c.mvp0 + 2) >> 2, c.fmin0, c.fmax0), pmy = clampi((c.mvp1 + 2) >> 2, c.fmin1, c.fmax1);
    bool done = false;
    int range = c.me_range;
    if (MODE == 1) {
        bx = 0; by = 0; bcost = 1 << 28;
        const int sx = cnd == 0 ? pmx : cnd == 2 ? clampi((c.col_x + 2) >> 2, c.fmin0, c.fmax0) : clampi(0, c.fmin0, c.fmax0);
        const int sy = cnd == 0 ? pmy : cnd == 2 ? clampi((c.col_y + 2) >> 2, c.fmin1, c.fmax1) : clampi(0, c.fmin1, c.fmax1);
        step(sx, sy, cnd < 2 || (cnd == 2 && c.has_col));
    } else {
    const int ucost1 = bcost;
    step(pmx + d1x, pmy + d1y, true);
    if (pmx | pmy) step(d1x, d1y, true);
    const int ucost2 = bcost;
    { const bool en = (bx | by) && ((bx - pmx) | (by - pmy)); if (__any(en)) step(bx + d1x, by + d1y, en); }
    int cross_start = bcost == ucost2 ? 3 : 1;
    const int omx = bx, omy = by;                       // x264 keeps the cross centred here
    // uneven cross: +-i along x for i = start, start+2, .. < xmax, then along y; four candidates (+i, -i, +(i+2), -(i+2)) per pass
    auto cross = [&](int start, int xmax, int ymax, bool en) {
        for (int t = 0; __any(en && start + 4 * t < xmax); t++) {
            const int i0 = start + 4 * t + 2 * (cnd >> 1), mx = omx + ((cnd & 1) ? -i0 : i0);
            step(mx, omy, en && i0 < xmax && ((cnd & 1) ? mx >= c.fmin0 : mx <= c.fmax0));
        }
        for (int t = 0; __any(en && start + 4 * t < ymax); t++) {
            const int i0 = start + 4 * t + 2 * (cnd >> 1), my = omy + ((cnd & 1) ? -i0 : i0);
            step(omx, my, en && i0 < ymax && ((cnd & 1) ? my >= c.fmin1 : my <= c.fmax1));
        }
    };
    const bool et = bcost == ucost2 && UMH_TH(2000);
    if (__any(et)) {                                    // early termination: small octagon
        step(omx + (cnd == 0 ? 0 : cnd == 1 ? -1 : cnd == 2 ? 1 : -2), omy + (cnd == 0 ? -2 : cnd == 3 ? 0 : -1), et);     // (0,-2) (-1,-1) (1,-1) (-2,0)
        step(omx + (cnd == 0 ? 2 : cnd == 1 ? -1 : cnd == 2 ? 1 : 0), omy + (cnd == 0 ? 0 : cnd == 3 ? 2 : 1), et);       // (2,0) (-1,1) (1,1) (0,2)
        done = et && bcost == ucost1 && UMH_TH(500);
        const bool et2 = et && !done && bcost == ucost2;
        if (__any(et2)) {
            const int r1 = (c.me_range >> 1) | 1;
            cross(3, r1, r1, et2);
            step(omx + (cnd == 0 ? -1 : cnd == 1 ? 1 : cnd == 2 ? -2 : 2), omy + (cnd < 2 ? -2 : -1), et2);                // (-1,-2) (1,-2) (-2,-1) (2,-1)
            step(omx + (cnd == 0 ? -2 : cnd == 1 ? 2 : cnd == 2 ? -1 : 1), omy + (cnd < 2 ? 1 : 2), et2);                  // (-2,1) (2,1) (-1,2) (1,2)
            done = done || (et2 && bcost == ucost2);
            if (et2) cross_start = r1 + 2;
        }
    }
    if (__any(!done)) {
        const bool live = !done;
        // adaptive search range: SAD level x disagreement of the predictors (x264 range_mul)
        const int sad_ctx = UMH_TH(1000) ? 0 : UMH_TH(2000) ? 1 : UMH_TH(4000) ? 2 : 3;
        const int mvd_ctx = mvd < 10 ? 0 : mvd < 20 ? 1 : mvd < 40 ? 2 : 3;
        range = (range * (int)((0x6544544444434433ull >> (4 * (mvd_ctx * 4 + sad_ctx))) & 15)) >> 2;
        cross(cross_start, range, range >> 1, live);
        step(omx + ((cnd & 2) ? 2 : -2), omy + ((cnd & 1) ? 2 : -2), live);                                                  // (-2,-2) (-2,2) (2,-2) (2,2)
        // 16-point hexagon grid rings around the best so far, radius 4*i
        const int hx = bx, hy = by;
        for (int i = 1; __any(live && (i == 1 || i <= (range >> 2))); i++) {          // do .. while (++i <= range >> 2)
            const bool en = live && (i == 1 || i <= (range >> 2));
            for (int ps = 0; ps < 4; ps++) {
                const int j = 4 * ps + cnd, mx = hx + umh_hex4_dx(j) * i, my = hy + umh_hex4_dy(j) * i;
                step(mx, my, en && inrange(mx, my));
            }
        }
        done = done || !inrange(bx, by);
    }
    }
#undef UMH_TH
    // hexagon (radius 2, up to range/2 - 1 moves) + square refine of the partitions that are still searching (x264 me_hex2)
    if (__any(!done)) {
        bool running = !done;
        int dir = 0;
        {
            const int cx0 = bx, cy0 = by;
            const int q1 = step(cx0 + hex_dx(1 + cnd), cy0 + hex_dy(1 + cnd), running);
            const int q2 = step(cx0 + hex_dx(5 + (cnd & 1)), cy0 + hex_dy(5 + (cnd & 1)), running && cnd < 2);
            const int kw = q2 >= 0 ? 5 + q2 : q1 >= 0 ? 1 + q1 : 0;
            running = running && kw != 0;
            dir = kw - 1;
        }
        for (int n = 0; ; n++) {
            running = running && n < (range >> 1) - 1 && inrange(bx, by);
            if (!__any(running)) break;
            const int cc = cnd < 3 ? cnd : 0, cx0 = bx, cy0 = by, d0 = running ? dir : 0;
            const int q = step(cx0 + hex_dx(d0 + cc), cy0 + hex_dy(d0 + cc), running && cnd < 3);
            if (running) {
                if (q < 0) running = false;
                else { dir += q - 1; dir = dir < 0 ? 5 : dir > 5 ? 0 : dir; }
            }
        }
        const int cx0 = bx, cy0 = by;
        step(cx0 + sq_dx(1 + cnd), cy0 + sq_dy(1 + cnd), !done);
        step(cx0 + sq_dx(5 + cnd), cy0 + sq_dy(5 + cnd), !done);
    }
}

// SHAPE 3: four 8x8 partitions (one per DPP row, four lanes per candidate); 1: two 16x8; 2: two 8x16 (32 lanes per partition,
// eight lanes per candidate).  A lane holds two 8-pixel segments of its partition: rows (2s, 2s+1) for the 8-wide shapes,
// the left and right half of row s for 16x8.
template <int M, bool UMH>
__device__ int search_parts(const PartCtx &c, const int SHAPE, int c0x, int c0y, int &out_mx, int &out_my)
{
    const int GL = SHAPE == 3 ? 4 : 8;             // SHAPE is wave-uniform: one copy of the code serves the three shapes
    const int lane = relane(c.lane), part = SHAPE == 3 ? lane >> 4 : lane >> 5, cnd = SHAPE == 3 ? (lane >> 2) & 3 : (lane >> 3) & 3, sr = lane & (GL - 1);
    const int ox = SHAPE == 3 ? (part & 1) * 8 : SHAPE == 2 ? part * 8 : 0, oy = SHAPE == 3 ? (part >> 1) * 8 : SHAPE == 1 ? part * 8 : 0;
    const int y0 = oy + (SHAPE == 1 ? sr : 2 * sr), y1 = SHAPE == 1 ? y0 : y0 + 1, x1 = SHAPE == 1 ? 8 : 0;
    uint32_t e0[2], e1[2];
    { const uint2 a = *(const uint2 *)(c.fenc + (size_t)y0 * c.fs + ox), b = *(const uint2 *)(c.fenc + (size_t)y1 * c.fs + ox + x1);
      e0[0] = a.x; e0[1] = a.y; e1[0] = b.x; e1[1] = b.y; }
    // chroma-ME: lane = row (sr & 3) of a 4x4 chroma block of the partition (16x8: blocks side by side, 8x16: stacked)
    const int ccx = (ox >> 1) + (SHAPE == 1 ? (sr >> 2) * 4 : 0), ccy = (oy >> 1) + (SHAPE == 2 ? sr : sr & 3);
    uint32_t ce0 = 0, ce1 = 0;
    if (c.chroma_me) { const uint2 v = *(const uint2 *)(c.fuv + (size_t)ccy * c.fs + 2 * ccx); ce0 = v.x; ce1 = v.y; }
    const int wb0 = (c.py + y0 - c.wy0) * WIN_STRIDE + (c.px + ox - c.wx0), wb1 = (c.py + y1 - c.wy0) * WIN_STRIDE + (c.px + ox + x1 - c.wx0);
    auto gsum = [&](int v) { v = quad_sum(v); const int w = xor4(v); return GL == 8 ? v + w : v; };  // over the candidate's lanes
    auto cmin = [&](unsigned k) {                                                                        // over the four candidates of a partition
        if (SHAPE == 3) return row16_min_u32(k);
        unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x128, 0xf, 0xf, true);           // row_ror:8: the other group of this DPP row
        k = t < k ? t : k;
        t = (unsigned)__shfl_xor((int)k, 16);
        return t < k ? t : k;
    };
    // candidate cost (uniform over the candidate's lanes): SAD of the partition at full-pel (fx,fy) + mv cost
#define FPC(fx, fy) (gsum(sad8_lds(c.win, wb0 + (fy) * WIN_STRIDE + (fx), e0[0], e0[1]) + sad8_lds(c.win, wb1 + (fy) * WIN_STRIDE + (fx), e1[0], e1[1])) + \
                     pc_mvcost<UMH>(c, (fx) * 4, (fy) * 4))
    int bx = c0x, by = c0y, bcost;
    if (UMH && c.la_mode) {
        bcost = 0;
        umh_fullpel(c, SHAPE, 1, 0, bx, by, bcost);             // lookahead: own start candidates, hexagon + square
    } else if (UMH) {
        {   // start candidate = the 16x16 vector (every group costs it)
            const long o = (long)by * c.rs + bx;
            const uint8_t *g0 = c.p00 + (long)(c.py + y0) * c.rs + c.px + ox, *g1 = c.p00 + (long)(c.py + y1) * c.rs + c.px + ox + x1;
            bcost = gsum(sad8_global(g0 + o, e0[0], e0[1]) + sad8_global(g1 + o, e1[0], e1[1])) + pc_mvcost<UMH>(c, bx * 4, by * 4);
        }
        umh_fullpel(c, SHAPE, 0, abs(c.mvp0 - 4 * c0x) + abs(c.mvp1 - 4 * c0y), bx, by, bcost);
    } else if (c.me_method == 3) {
        // exhaustive search (see the 16x16 path): the rectangle is the same for every partition (common start and limits)
        bcost = FPC(bx, by);
        const int rr = c.me_range;
        const int min_x = max(bx - rr, c.fmin0), min_y = max(by - rr, c.fmin1), max_x = min(bx + rr, c.fmax0), max_y = min(by + rr, c.fmax1);
        const int width = (max_x - min_x + 3) & ~3;
        unsigned kmin = 0xffffffffu;
        for (int my = min_y; my <= max_y; my++)
            for (int x4 = 0; x4 < width; x4 += 4) {
                const int mx = min_x + x4 + cnd;
                kmin = min(kmin, ((unsigned)FPC(mx, my) << 11) | (unsigned)((my - min_y) * width + x4 + cnd));
            }
        kmin = cmin(kmin);
        if (kmin != 0xffffffffu && (int)(kmin >> 11) < bcost) {
            const int idx = (int)(kmin & 2047);
            bcost = (int)(kmin >> 11); bx = min_x + idx % width; by = min_y + idx / width;
        }
    } else if (c.me_method == 0) {
        bcost = FPC(bx, by);
        bool running = true;
        for (int it = c.me_range; it > 0; it--) {
            if (!__any(running)) break;
            const int cx = bx + (cnd == 2 ? -1 : cnd == 3 ? 1 : 0), cy = by + (cnd == 0 ? -1 : cnd == 1 ? 1 : 0);
            const unsigned kk = cmin(((unsigned)FPC(cx, cy) << 2) | (unsigned)cnd);
            if (running) {
                if ((int)(kk >> 2) < bcost) {
                    const int q = kk & 3;
                    bcost = (int)(kk >> 2);
                    bx += q == 2 ? -1 : q == 3 ? 1 : 0; by += q == 0 ? -1 : q == 1 ? 1 : 0;
                    running = bx >= c.fmin0 && bx <= c.fmax0 && by >= c.fmin1 && by <= c.fmax1;
                } else running = false;
            }
        }
    } else {
        unsigned key = (unsigned)FPC(bx, by) << 3;
        {   // first ring: hex2[1..6], tags 2..7, in two passes of four candidates
            int i = 1 + cnd;
            unsigned kk = ((unsigned)FPC(bx + hex_dx(i), by + hex_dy(i)) << 3) | (unsigned)(i + 1);
            key = min(key, cmin(kk));
            i = 5 + (cnd & 1);
            kk = ((unsigned)FPC(bx + hex_dx(i), by + hex_dy(i)) << 3) | (unsigned)(i + 1);
            if (cnd >= 2) kk = 0xffffffffu;
            key = min(key, cmin(kk));
        }
        bool running = (key & 7) != 0;
        int dir = running ? (int)(key & 7) - 2 : 0;
        if (running) { bx += hex_dx(dir + 1); by += hex_dy(dir + 1); }
        for (int it = (c.me_range >> 1) - 1; it > 0; it--) {
            if (!__any(running)) break;
            running = running && bx >= c.fmin0 && bx <= c.fmax0 && by >= c.fmin1 && by <= c.fmax1;
            const int cc = cnd < 3 ? cnd : 0;
            unsigned kk = ((unsigned)FPC(bx + hex_dx(dir + cc), by + hex_dy(dir + cc)) << 3) | (unsigned)(cc + 1);
            if (cnd >= 3) kk = 0xffffffffu;
            const unsigned k2 = min(key & ~7u, cmin(kk));
            if (running) {
                key = k2;
                if (!(key & 7)) running = false;
                else {
                    dir += (int)(key & 7) - 2;
                    dir = dir < 0 ? 5 : dir > 5 ? 0 : dir;
                    bx += hex_dx(dir + 1); by += hex_dy(dir + 1);
                }
            }
        }
        bcost = (int)(key >> 3);
        {   // square refine: square1[1..8], first strictly-better candidate in order wins
            unsigned sk = (unsigned)bcost << 4;
            int q = 1 + cnd;
            sk = min(sk, cmin(((unsigned)FPC(bx + sq_dx(q), by + sq_dy(q)) << 4) | (unsigned)q));
            q = 5 + cnd;
            sk = min(sk, cmin(((unsigned)FPC(bx + sq_dx(q), by + sq_dy(q)) << 4) | (unsigned)q));
            const int bd = sk & 15;
            bcost = (int)(sk >> 4);
            if (bd) { bx += sq_dx(bd); by += sq_dy(bd); }
        }
    }
#undef FPC
    int mx = bx * 4, my = by * 4;
    if (c.hp_it > 0) {
        const s16x2 sg1 = pk_sign(lane & 1), sg2 = pk_sign(lane & 2);
        const int PW = SHAPE == 1 ? 16 : 8, PH = SHAPE == 2 ? 16 : 8, L = SHAPE == 3 ? 16 : 32;
        const int rwl = SubGeo<M>::rwl(PW), rh = SubGeo<M>::rh(PH), ncol = SubGeo<M>::ncol(PW), sn = rh << rwl;
        const int sx0 = (c.px + ox + bx - M) & ~3, sy0 = c.py + oy + by - M;
        uint32_t *sb = c.sub + part * 4 * sn;
        __builtin_amdgcn_wave_barrier();
        sub_stage<M>(sb, c.p00, c.pb, c.rs, sx0, sy0, rwl, rh, ncol, lane & (L - 1), L);
        // chroma neighbourhood of this partition (chroma-ME)
        const int cwc = PW >> 1, chc = PH >> 1, cndw = CSubGeo<M>::ndw(cwc), cnr = CSubGeo<M>::rows(chc);
        const int cx0c = ((c.px >> 1) + (ox >> 1) + (bx >> 1) - CSubGeo<M>::MG) & ~1, cy0c = (c.py >> 1) + (oy >> 1) + (by >> 1) - CSubGeo<M>::MG;
        uint32_t *cb = c.csub + part * (cndw * cnr);
        if (c.chroma_me) chroma_stage(cb, c.cref, c.rs, cx0c, cy0c, cndw, cnr, lane & (L - 1), L);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int X0 = c.px + ox, Y0 = c.py + y0, X1 = X0 + x1, Y1 = c.py + y1;
        // SATD of the partition for the prediction segments p0 / p1 (half share per lane, summed over the candidate's lanes)
        auto satd_part = [&](const uint32_t p0[2], const uint32_t p1[2]) {
            if (SHAPE == 1) {
                const uint32_t e[4] = { e0[0], e0[1], e1[0], e1[1] }, p[4] = { p0[0], p0[1], p1[0], p1[1] };
                return gsum(satd16x4_half_pk(e, p, sg1, sg2));
            }
            return gsum(satd8x8_2rows_half(e0, e1, p0, p1, sg1));
        };
        bool hp_run = true;
        for (int it = c.hp_it; it > 0; it--) {
            if (!__any(hp_run)) break;
            const int cx = mx + (cnd == 2 ? -2 : cnd == 3 ? 2 : 0), cy = my + (cnd == 0 ? -2 : cnd == 1 ? 2 : 0);
            uint32_t p0[2], p1[2];
            sub_row8(sb, sn, rwl, sx0, sy0, X0, Y0, cx, cy, p0);
            sub_row8(sb, sn, rwl, sx0, sy0, X1, Y1, cx, cy, p1);
            unsigned sd = __builtin_amdgcn_sad_u8(p0[0], e0[0], 0u);
            sd = __builtin_amdgcn_sad_u8(p0[1], e0[1], sd); sd = __builtin_amdgcn_sad_u8(p1[0], e1[0], sd); sd = __builtin_amdgcn_sad_u8(p1[1], e1[1], sd);
            const unsigned kk = cmin(((unsigned)(gsum((int)sd) + pc_mvcost<UMH>(c, cx, cy)) << 2) | (unsigned)cnd);
            if (hp_run && (int)(kk >> 2) < bcost) {
                const int b = kk & 3;
                bcost = (int)(kk >> 2);
                mx += b == 2 ? -2 : b == 3 ? 2 : 0; my += b == 0 ? -2 : b == 1 ? 2 : 0;
            } else hp_run = false;
        }
        {   // SATD at the best half-pel position (every candidate group computes the same value)
            uint32_t p0[2], p1[2];
            sub_row8(sb, sn, rwl, sx0, sy0, X0, Y0, mx, my, p0);
            sub_row8(sb, sn, rwl, sx0, sy0, X1, Y1, mx, my, p1);
            bcost = satd_part(p0, p1) + pc_mvcost<UMH>(c, mx, my);
            if (c.chroma_me) bcost += gsum(chroma_me_lds(cb, cndw, cx0c, cy0c, (c.px >> 1) + ccx, (c.py >> 1) + ccy, mx, my, ce0, ce1, sg1, sg2));
        }
        int bdir = -1;
        bool qp_run = true;
        for (int it = c.qp_it; it > 0; it--) {
            if (!__any(qp_run)) break;
            qp_run = qp_run && !(my <= c.smin1 || my >= c.smax1 || mx <= c.smin0 || mx >= c.smax0);
            const int cx = mx + (cnd == 2 ? -1 : cnd == 3 ? 1 : 0), cy = my + (cnd == 0 ? -1 : cnd == 1 ? 1 : 0);
            uint32_t p0[2], p1[2];
            sub_row8(sb, sn, rwl, sx0, sy0, X0, Y0, cx, cy, p0);
            sub_row8(sb, sn, rwl, sx0, sy0, X1, Y1, cx, cy, p1);
            int cst = satd_part(p0, p1) + pc_mvcost<UMH>(c, cx, cy);
            // a candidate whose luma cost is not below the best cost cannot win; chroma only when some partition still can improve
            if (c.chroma_me && __any(qp_run && cst < bcost && (cnd ^ 1) != bdir))
                cst += gsum(chroma_me_lds(cb, cndw, cx0c, cy0c, (c.px >> 1) + ccx, (c.py >> 1) + ccy, cx, cy, ce0, ce1, sg1, sg2));
            unsigned kk = ((unsigned)cst << 2) | (unsigned)cnd;
            if ((cnd ^ 1) == bdir) kk = 0xffffffffu;
            kk = cmin(kk);
            if (qp_run && (int)(kk >> 2) < bcost) {
                bcost = (int)(kk >> 2);
                bdir = kk & 3;
                mx += bdir == 2 ? -1 : bdir == 3 ? 1 : 0; my += bdir == 0 ? -1 : bdir == 1 ? 1 : 0;
            } else qp_run = false;
        }
    }
    if (SHAPE == 2) {       // 8x16: partition k owns lanes 32k.., but callers index vectors by 8x8 block (lane >> 4): blocks 1, 3 are partition 1
        const int src = ((lane >> 4) & 1) * 32;
        out_mx = __shfl(mx, src); out_my = __shfl(my, src);
    } else { out_mx = mx; out_my = my; }
    return bcost;
}

}  // namespace x264gpu

// mc.hip.h — motion-compensated sample fetch for gfx950 (oracle/mc.c: x264o_mc_luma / x264o_mc_chroma;
// normative H.264 8.4.2.2).  Luma quarter-pel samples come from the four half-pel planes (full, H, V,
// HV) resident in HBM: at most two unaligned dword loads + one packed rounding average per 4 pixels.
#pragma once
#include "dsp.hip.h"

namespace x264gpu {

constexpr uint32_t pack_2bit_table(const int (&t)[16])
{
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r |= (uint32_t)t[i] << (2 * i);
    return r;
}
constexpr int kQpelPlane0[16] = { 0, 1, 1, 1, 0, 1, 1, 1, 2, 3, 3, 3, 0, 1, 1, 1 };
constexpr int kQpelPlane1[16] = { 0, 0, 1, 0, 2, 2, 3, 2, 2, 2, 3, 2, 2, 2, 3, 2 };
constexpr uint32_t kQpelPlane0Packed = pack_2bit_table(kQpelPlane0);
constexpr uint32_t kQpelPlane1Packed = pack_2bit_table(kQpelPlane1);

// 4 luma pixels at integer position (x..x+3, y) displaced by the quarter-pel vector (mvx,mvy).
// p00 addresses sample (0,0) of plane 0; plane k is k*plane_bytes further.
__device__ __forceinline__ uint32_t mc_luma_row4(const uint8_t *__restrict__ p00, size_t plane_bytes, int stride,
                                                 int x, int y, int mvx, int mvy)
{
    const int idx = ((mvy & 3) << 2) | (mvx & 3);
    const long base = (long)(y + (mvy >> 2)) * stride + x + (mvx >> 2);
    const int pl0 = (kQpelPlane0Packed >> (2 * idx)) & 3, pl1 = (kQpelPlane1Packed >> (2 * idx)) & 3;
    // (both loads go out together, whether or not the position needs the second plane: a load behind a per-lane branch waits for the first one's
    //  latency before it is even issued; where it is not needed it reads a valid sample of a resident plane and is dropped)
    const uint32_t a = load_u32_unaligned(p00 + pl0 * plane_bytes + base + ((mvy & 3) == 3 ? stride : 0));
    const uint32_t b = load_u32_unaligned(p00 + pl1 * plane_bytes + base + ((mvx & 3) == 3 ? 1 : 0));
    return (idx & 5) ? avg4_u8(a, b) : a;
}

// 4 chroma pixels (U and V) at chroma position (x..x+3, y) from a padded NV12 plane; mv in 1/8 chroma pel
__device__ __forceinline__ void mc_chroma_row4(const uint8_t *__restrict__ nv12, int stride, int x, int y,
                                               int mvx, int mvy, uint32_t &u, uint32_t &v)
{
    const int dx = mvx & 7, dy = mvy & 7;
    const int cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
    const uint8_t *s = nv12 + (long)(y + (mvy >> 3)) * stride + 2 * (x + (mvx >> 3));
    uint32_t r0[3], r1[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { r0[i] = load_u32_unaligned(s + 4 * i); r1[i] = load_u32_unaligned(s + stride + 4 * i); }
    // bytes: U0 V0 U1 V1 | U2 V2 U3 V3 | U4 V4 .. ..
    int t0[10], t1[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { t0[i] = (r0[i >> 2] >> (8 * (i & 3))) & 0xff; t1[i] = (r1[i >> 2] >> (8 * (i & 3))) & 0xff; }
    int pu[4], pv[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        pu[i] = (cA * t0[2 * i] + cB * t0[2 * i + 2] + cC * t1[2 * i] + cD * t1[2 * i + 2] + 32) >> 6;
        pv[i] = (cA * t0[2 * i + 1] + cB * t0[2 * i + 3] + cC * t1[2 * i + 1] + cD * t1[2 * i + 3] + 32) >> 6;
    }
    u = pack4(pu);
    v = pack4(pv);
}


// bi-prediction of four packed samples (x264 pixel_avg_weight_wxh): weight1 == 32 is the rounding average
__device__ __forceinline__ uint32_t avg_weight4_u8(uint32_t a, uint32_t b, int w1)
{
    if (w1 == 32) return avg4_u8(a, b);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = ((int)((a >> (8 * i)) & 255) * w1 + (int)((b >> (8 * i)) & 255) * (64 - w1) + 32) >> 6;
        o |= (uint32_t)min(max(v, 0), 255) << (8 * i);
    }
    return o;
}
// explicit weighted prediction of four packed samples (x264 mc_weight)
__device__ __forceinline__ uint32_t weight4_u8(uint32_t s, int scale, int denom, int offset)
{
    uint32_t o = 0;
    const int rnd = denom >= 1 ? 1 << (denom - 1) : 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int v = (((int)((s >> (8 * i)) & 255) * scale + rnd) >> denom) + offset;
        o |= (uint32_t)min(max(v, 0), 255) << (8 * i);
    }
    return o;
}

}  // namespace x264gpu

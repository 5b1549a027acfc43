// frame_kernels.hip — whole-frame streaming kernels: half-pel plane filter + border expansion (A4)
// and the lowres pyramid (A1).  HBM-bound streaming kernels.
#include "common.hip.h"

using namespace x264gpu;

namespace {

constexpr int HP_ROWS = 16;                 // output rows per thread (5 more source rows are read per strip)

// ---- half-pel planes (A4).  A thread owns 4 columns x HP_ROWS rows and slides a 6-row window down the picture held in
// registers: per row it fetches 12 source bytes (three aligned dwords, x0-4 .. x0+7; HBM sees every byte once, the overlap
// between neighbouring lanes is L1/L2 traffic), keeps them unpacked as packed 16-bit pairs, and produces
//   V  = clip((6-tap over the 6 rows + 16) >> 5)            packed 16-bit arithmetic (|value| <= 10710)
//   H  = clip((6-tap along the current row + 16) >> 5)      packed 16-bit, neighbours via v_alignbit on the pair registers
//   HV = clip((6-tap along the row of the un-rounded vertical sums + 512) >> 10)   32-bit
// No LDS, no barriers; source reads clamp to the picture (== edge replication); the full-pel border is written here too.
// Restates oracle/mc.c x264o_frame_filter bit-exactly.
struct HpRow { s16x2 e[3], o[3]; };          // 12 columns x0-4 .. x0+7: e[d] = (c[4d], c[4d+2]), o[d] = (c[4d+1], c[4d+3])

__device__ __forceinline__ HpRow hp_load_row(const uint8_t *__restrict__ src, int stride, int w, int h, int x0, int y, bool interior)
{
    const int yc = min(max(y, 0), h - 1);
    const uint8_t *r = src + (size_t)yc * stride;
    uint32_t d[3];
    if (interior) {
        const uint32_t *p = (const uint32_t *)(r + x0 - 4);
        d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            uint32_t v = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) v |= (uint32_t)r[min(max(x0 - 4 + 4 * k + i, 0), w - 1)] << (8 * i);
            d[k] = v;
        }
    }
    HpRow o;
#pragma unroll
    for (int k = 0; k < 3; k++) { o.e[k] = pk_even(d[k]); o.o[k] = pk_odd(d[k]); }
    return o;
}
__device__ __forceinline__ s16x2 hp_tap6(s16x2 a, s16x2 b, s16x2 c, s16x2 d, s16x2 e, s16x2 f)
{
    const s16x2 k5 = as_s16x2(0x00050005u), k20 = as_s16x2(0x00140014u);
    return (a + f) - (b + e) * k5 + (c + d) * k20;
}
__device__ __forceinline__ s16x2 hp_hi_lo(s16x2 hi, s16x2 lo) { return as_s16x2(__builtin_amdgcn_alignbit(as_u32(hi), as_u32(lo), 16)); }   // (lo.hi, hi.lo)
// (v + 16) >> 5 clipped to 0..255, pairs (p0,p2) / (p1,p3) -> 4 packed bytes
__device__ __forceinline__ uint32_t hp_round_pack(s16x2 v02, s16x2 v13)
{
    const s16x2 r = as_s16x2(0x00100010u), z = as_s16x2(0u), m = as_s16x2(0x00ff00ffu);
    const s16x2 a = __builtin_elementwise_min(__builtin_elementwise_max((v02 + r) >> 5, z), m);
    const s16x2 b = __builtin_elementwise_min(__builtin_elementwise_max((v13 + r) >> 5, z), m);
    return as_u32(a) | (as_u32(b) << 8);
}

__global__ __launch_bounds__(256) void k_hpel_filter(uint8_t *__restrict__ planes0, size_t plane_bytes, int stride,
                                                     int w, int h, int pad, size_t batch_bytes)
{
    uint8_t *__restrict__ planes = planes0 + (size_t)blockIdx.z * batch_bytes;   // blockIdx.z = stream
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4 - pad, ys = blockIdx.y * HP_ROWS - pad;  // picture coords
    if (x0 >= w + pad) return;
    const uint8_t *src = planes + (size_t)pad * stride + pad;
    const bool interior = x0 - 4 >= 0 && x0 + 8 <= w;
    const bool x_out = x0 + 3 < 0 || x0 >= w;            // w and pad are multiples of 4: a 4-column cell never straddles the edge
    HpRow r0 = hp_load_row(src, stride, w, h, x0, ys - 2, interior), r1 = hp_load_row(src, stride, w, h, x0, ys - 1, interior);
    HpRow r2 = hp_load_row(src, stride, w, h, x0, ys, interior), r3 = hp_load_row(src, stride, w, h, x0, ys + 1, interior);
    HpRow r4 = hp_load_row(src, stride, w, h, x0, ys + 2, interior);
#pragma unroll
    for (int j = 0; j < HP_ROWS; j++) {
        const int Y = ys + j;
        if (Y >= h + pad) break;
        const HpRow r5 = hp_load_row(src, stride, w, h, x0, Y + 3, interior);
        // un-rounded vertical sums for the 12 columns
        s16x2 ve[3], vo[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            ve[d] = hp_tap6(r0.e[d], r1.e[d], r2.e[d], r3.e[d], r4.e[d], r5.e[d]);
            vo[d] = hp_tap6(r0.o[d], r1.o[d], r2.o[d], r3.o[d], r4.o[d], r5.o[d]);
        }
        const size_t off = (size_t)(Y + pad) * stride + (x0 + pad);
        // V: own columns c[4..7] = (ve[1].lo, vo[1].lo, ve[1].hi, vo[1].hi)
        *(uint32_t *)(planes + 2 * plane_bytes + off) = hp_round_pack(ve[1], vo[1]);
        // H: pixel i taps c[i+2 .. i+7] of the current row (r2); pairs (px0,px2) and (px1,px3)
        {
            const s16x2 c24 = hp_hi_lo(r2.e[1], r2.e[0]), c35 = hp_hi_lo(r2.o[1], r2.o[0]), c68 = hp_hi_lo(r2.e[2], r2.e[1]), c79 = hp_hi_lo(r2.o[2], r2.o[1]);
            const s16x2 h02 = hp_tap6(c24, c35, r2.e[1], r2.o[1], c68, c79);
            const s16x2 h13 = hp_tap6(c35, r2.e[1], r2.o[1], c68, c79, r2.e[2]);
            *(uint32_t *)(planes + plane_bytes + off) = hp_round_pack(h02, h13);
        }
        // HV: the same taps over the vertical sums, in 32 bits
        {
            int c[9];                                    // columns k = 2 .. 10
            c[0] = ve[0].y; c[1] = vo[0].y; c[2] = ve[1].x; c[3] = vo[1].x; c[4] = ve[1].y; c[5] = vo[1].y; c[6] = ve[2].x; c[7] = vo[2].x; c[8] = ve[2].y;
            int p[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                p[i] = clip_u8((c[i] - 5 * c[i + 1] + 20 * c[i + 2] + 20 * c[i + 3] - 5 * c[i + 4] + c[i + 5] + 512) >> 10);
            *(uint32_t *)(planes + 3 * plane_bytes + off) = pack4(p);
        }
        // full-pel border: cells outside the picture take the clamped source (own columns of the current row)
        if (x_out || Y < 0 || Y >= h) *(uint32_t *)(planes + off) = as_u32(r2.e[1]) | (as_u32(r2.o[1]) << 8);
        r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5;
    }
}

__device__ __forceinline__ int avg4r(int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; }

__global__ __launch_bounds__(256) void k_lowres(const uint8_t *__restrict__ src, int ss, int w, int h,
                                                uint8_t *__restrict__ dst, size_t plane_bytes, int ds)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w / 2 || y >= h / 2) return;
    int p[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
            p[j][i] = src[(size_t)min(2 * y + j, h - 1) * ss + min(2 * x + i, w - 1)];
    size_t o = (size_t)y * ds + x;
    dst[o] = (uint8_t)avg4r(p[0][0], p[1][0], p[0][1], p[1][1]);
    dst[plane_bytes + o] = (uint8_t)avg4r(p[0][1], p[1][1], p[0][2], p[1][2]);
    dst[2 * plane_bytes + o] = (uint8_t)avg4r(p[1][0], p[2][0], p[1][1], p[2][1]);
    dst[3 * plane_bytes + o] = (uint8_t)avg4r(p[1][1], p[2][1], p[1][2], p[2][2]);
}

}  // namespace

namespace x264gpu {
int launch_hpel_filter(uint8_t *planes, size_t plane_bytes, int stride, int w, int h, int pad, int batch,
                       size_t batch_bytes, hipStream_t st)
{
    dim3 grid((w + 2 * pad + 1023) / 1024, (h + 2 * pad + HP_ROWS - 1) / HP_ROWS, batch);
    hipLaunchKernelGGL(k_hpel_filter, grid, dim3(256), 0, st, planes, plane_bytes, stride, w, h, pad, batch_bytes);
    return 0;
}
}  // namespace x264gpu

extern "C" {

int x264gpu_hpel_filter(uint8_t *d_planes, size_t plane_bytes, int stride, int w, int h, int pad, void *stream)
{
    ARG_TRY(d_planes && w > 0 && h > 0 && pad >= 8 && (stride % 4) == 0 && stride >= w + 2 * pad && ((w + 2 * pad) % 4) == 0);
    ARG_TRY(plane_bytes >= (size_t)stride * (h + 2 * pad) && (plane_bytes % 4) == 0 && (pad % 4) == 0);
    launch_hpel_filter(d_planes, plane_bytes, stride, w, h, pad, 1, 0, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_lowres(const uint8_t *d_src, int ss, int w, int h, uint8_t *d_dst, size_t plane_bytes, int ds, void *stream)
{
    ARG_TRY(d_src && d_dst && w >= 2 && h >= 2 && ds >= w / 2 && plane_bytes >= (size_t)ds * (h / 2));
    dim3 grid((w / 2 + 63) / 64, (h / 2 + 3) / 4);
    hipLaunchKernelGGL(k_lowres, grid, dim3(256), 0, (hipStream_t)stream, d_src, ss, w, h, d_dst, plane_bytes, ds);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

}  // extern "C"

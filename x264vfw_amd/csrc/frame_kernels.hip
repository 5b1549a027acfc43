// frame_kernels.hip — whole-frame streaming kernels: half-pel plane filter + border expansion (A4)
// and the lowres pyramid (A1).  HBM-bound: each source byte is fetched once per tile (+halo) into LDS.
#include "common.cuh"

using namespace x264gpu;

namespace {

constexpr int HP_TW = 64, HP_TH = 16;       // output tile
constexpr int HP_SW = 72, HP_SH = HP_TH + 5; // source tile incl. 6-tap halo (cols x-2..x+66, rows y-2..y+18)

__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

// planes: padded origin of plane 0; sample (0,0) at pad*stride+pad.  One 256-thread block per 64x16 tile
// of the PADDED plane; source reads are clamped to the picture (== edge replication, oracle/mc.c).
__global__ __launch_bounds__(256) void k_hpel_filter(uint8_t *__restrict__ planes0, size_t plane_bytes, int stride,
                                                     int w, int h, int pad, size_t batch_bytes)
{
    uint8_t *__restrict__ planes = planes0 + (size_t)blockIdx.z * batch_bytes;   // blockIdx.z = stream
    __shared__ uint8_t s[HP_SH][HP_SW];
    __shared__ int16_t vi[HP_TH][HP_SW];
    const int t = threadIdx.x;
    const int px0 = blockIdx.x * HP_TW - pad, py0 = blockIdx.y * HP_TH - pad;  // picture coords of tile origin
    const uint8_t *src = planes + (size_t)pad * stride + pad;
    for (int i = t; i < HP_SH * HP_SW; i += 256) {
        int r = i / HP_SW, c = i % HP_SW;
        int y = min(max(py0 - 2 + r, 0), h - 1), x = min(max(px0 - 2 + c, 0), w - 1);
        s[r][c] = src[(size_t)y * stride + x];
    }
    __syncthreads();
    for (int i = t; i < HP_TH * HP_SW; i += 256) {
        int r = i / HP_SW, c = i % HP_SW;
        vi[r][c] = (int16_t)tap6(s[r][c], s[r + 1][c], s[r + 2][c], s[r + 3][c], s[r + 4][c], s[r + 5][c]);
    }
    __syncthreads();
    const int ty = t >> 4, tx = (t & 15) * 4;
    const int X = px0 + tx, Y = py0 + ty;   // picture coords of this thread's first pixel
    if (X >= w + pad || Y >= h + pad) return;
    int ph[4], pv[4], pc[4], pf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int c = tx + i;
        pf[i] = s[ty + 2][c + 2];
        ph[i] = clip_u8((tap6(s[ty + 2][c], s[ty + 2][c + 1], s[ty + 2][c + 2], s[ty + 2][c + 3], s[ty + 2][c + 4], s[ty + 2][c + 5]) + 16) >> 5);
        pv[i] = clip_u8((vi[ty][c + 2] + 16) >> 5);
        pc[i] = clip_u8((tap6(vi[ty][c], vi[ty][c + 1], vi[ty][c + 2], vi[ty][c + 3], vi[ty][c + 4], vi[ty][c + 5]) + 512) >> 10);
    }
    size_t off = (size_t)(Y + pad) * stride + (X + pad);
    const bool inside = X >= 0 && X + 3 < w && Y >= 0 && Y < h;
    if (!inside) {
        // border of the full-pel plane: only bytes outside the picture are written
        if (X + 3 < 0 || X >= w || Y < 0 || Y >= h) *(uint32_t *)(planes + off) = pack4(pf);
        else
            for (int i = 0; i < 4; i++)
                if (X + i < 0 || X + i >= w) planes[off + i] = (uint8_t)pf[i];
    }
    *(uint32_t *)(planes + plane_bytes + off) = pack4(ph);
    *(uint32_t *)(planes + 2 * plane_bytes + off) = pack4(pv);
    *(uint32_t *)(planes + 3 * plane_bytes + off) = pack4(pc);
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
    dim3 grid((w + 2 * pad + HP_TW - 1) / HP_TW, (h + 2 * pad + HP_TH - 1) / HP_TH, batch);
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

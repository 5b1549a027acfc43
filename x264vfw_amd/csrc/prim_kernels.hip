// prim_kernels.hip — Tier-1 batch primitives of include/x264gpu.h ("checkasm" surface).
// Thin kernels around the wave-level device library (dsp.cuh, mc.cuh) that the frame pipeline uses,
// so each primitive is parity-tested against oracle/ with exactly the device code that ships.
#include "common.cuh"
#include "mc.cuh"

using namespace x264gpu;

// ------------------------------------------------------------------------------------------------
// pixel metrics: one wavefront per block pair; lane = (4x4 block, row) so quads are 4x4 blocks
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pixel_metric(int metric, const uint8_t *__restrict__ a,
                                                      const uint8_t *__restrict__ b, int n, int w, int h,
                                                      int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n) return;  // wave-uniform
    const int nbx = w >> 2, nblk = nbx * (h >> 2);
    const int blk = lane >> 2, j = lane & 3;
    int bx, by;
    if (w == 16 && h == 16) { bx = z_bx(blk); by = z_by(blk); }
    else { bx = blk % nbx; by = blk / nbx; }
    const bool valid = blk < nblk;
    uint32_t pa = 0, pb = 0;
    if (valid) {
        size_t off = (size_t)idx * w * h + (by * 4 + j) * w + bx * 4;
        pa = *(const uint32_t *)(a + off);
        pb = *(const uint32_t *)(b + off);
    }
    int va[4], vb[4], d[4];
    unpack4(pa, va); unpack4(pb, vb);
#pragma unroll
    for (int i = 0; i < 4; i++) d[i] = va[i] - vb[i];
    int r;
    if (metric == 0) r = wave_sum(sad4(pa, pb));
    else if (metric == 3) r = wave_sum(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    else if (metric == 1) r = wave_sum(satd_quad_partial(d, lane)) >> 1;
    else {
        // SA8D: 8x8 Hadamard = 2x2 butterfly across the four 4x4 Hadamards of an 8x8 (H8 = H2 (x) H4)
        hadamard4_quad(d, lane);
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int y = __shfl_xor(d[i], 4);
            int t = (lane & 4) ? y - d[i] : d[i] + y;
            y = __shfl_xor(t, 8);
            t = (lane & 8) ? y - t : t + y;
            s += abs(t);
        }
        r = (wave_sum(valid ? s : 0) + 2) >> 2;
    }
    if (lane == 0) out[idx] = r;
}

__global__ __launch_bounds__(256) void k_pixel_var(const uint8_t *__restrict__ a, int n, int w, int h,
                                                   uint64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int nseg = (w >> 2) * h;
    uint32_t p = 0;
    if (lane < nseg) p = *(const uint32_t *)(a + (size_t)idx * w * h + lane * 4);
    int v[4];
    unpack4(p, v);
    unsigned sum = wave_sum(v[0] + v[1] + v[2] + v[3]);
    unsigned sqr = wave_sum(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    if (lane == 0) out[idx] = sum + ((uint64_t)sqr << 32);
}

// ------------------------------------------------------------------------------------------------
// 4x4 residual pipeline: dct -> quant -> dequant -> idct, 16 blocks per wavefront
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dctq4x4(const uint8_t *__restrict__ enc, const uint8_t *__restrict__ pred,
                                                 int n, Q4 q, int16_t *__restrict__ coef,
                                                 int16_t *__restrict__ levels, uint8_t *__restrict__ recon)
{
    const int lane = threadIdx.x & 63;
    const int blk = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (lane >> 2);
    const int j = lane & 3;
    const bool valid = blk < n;
    uint32_t pe = 0, pp = 0;
    if (valid) {
        pe = *(const uint32_t *)(enc + (size_t)blk * 16 + j * 4);
        pp = *(const uint32_t *)(pred + (size_t)blk * 16 + j * 4);
    }
    int e[4], p[4], v[4];
    unpack4(pe, e); unpack4(pp, p);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
    dct4_quad(v, lane);
    if (coef && valid)
        *(short4 *)(coef + (size_t)blk * 16 + j * 4) = make_short4((short)v[0], (short)v[1], (short)v[2], (short)v[3]);
    quant4_row(v, q, j);
    if (levels && valid)
        *(short4 *)(levels + (size_t)blk * 16 + j * 4) = make_short4((short)v[0], (short)v[1], (short)v[2], (short)v[3]);
    dequant4_row(v, q, j);
    idct4_quad(v, lane);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] += p[i];
    if (recon && valid) *(uint32_t *)(recon + (size_t)blk * 16 + j * 4) = pack4_clip(v);
}

// ------------------------------------------------------------------------------------------------
// motion compensation batch kernels (one wavefront per block)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mc_luma(const uint8_t *__restrict__ p00, size_t plane_bytes, int stride,
                                                 const int32_t *__restrict__ xy, const int32_t *__restrict__ mv,
                                                 int n, int w, int h, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int nbx = w >> 2;
    if (lane >= nbx * h) return;
    const int x0 = (lane % nbx) * 4, y = lane / nbx;
    uint32_t v = mc_luma_row4(p00, plane_bytes, stride, xy[2 * idx] + x0, xy[2 * idx + 1] + y, mv[2 * idx], mv[2 * idx + 1]);
    *(uint32_t *)(out + (size_t)idx * w * h + y * w + x0) = v;
}

// bi-prediction / explicit weighting of already motion-compensated samples, four per thread (any block shape: the blocks are contiguous)
__global__ __launch_bounds__(256) void k_mc_avg(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n, int weight1, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = avg_weight4_u8(a[i], b[i], weight1);
}
__global__ __launch_bounds__(256) void k_mc_weight(const uint32_t *__restrict__ src, size_t n, int scale, int denom, int offset, uint32_t *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = weight4_u8(src[i], scale, denom, offset);
}

__global__ __launch_bounds__(256) void k_mc_chroma(const uint8_t *__restrict__ nv12, int stride,
                                                   const int32_t *__restrict__ xy, const int32_t *__restrict__ mv,
                                                   int n, int w, int h, uint8_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int nbx = w >> 2;
    if (lane >= nbx * h) return;
    const int x0 = (lane % nbx) * 4, y = lane / nbx;
    uint32_t u, v;
    mc_chroma_row4(nv12, stride, xy[2 * idx] + x0, xy[2 * idx + 1] + y, mv[2 * idx], mv[2 * idx + 1], u, v);
    uint8_t *o = out + (size_t)idx * w * h * 2;
    *(uint32_t *)(o + y * w + x0) = u;
    *(uint32_t *)(o + w * h + y * w + x0) = v;
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int x264gpu_pixel_metric(int metric, const uint8_t *d_a, const uint8_t *d_b, int n, int w, int h,
                         int32_t *d_out, void *stream)
{
    ARG_TRY(metric >= 0 && metric <= 3 && n >= 0 && d_a && d_b && d_out);
    ARG_TRY((w == 4 || w == 8 || w == 16) && (h == 4 || h == 8 || h == 16));
    ARG_TRY(metric != 2 || ((w == 8 && h == 8) || (w == 16 && h == 16)));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_pixel_metric, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, metric, d_a, d_b, n, w, h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_pixel_var(const uint8_t *d_a, int n, int w, int h, uint64_t *d_out, void *stream)
{
    ARG_TRY(n >= 0 && d_a && d_out && ((w == 16 && h == 16) || (w == 8 && h == 8) || (w == 8 && h == 16)));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_pixel_var, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_a, n, w, h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_dctq4x4(const uint8_t *d_enc, const uint8_t *d_pred, int n, int qp, int list, int16_t *d_coef,
                    int16_t *d_levels, uint8_t *d_recon, void *stream)
{
    ARG_TRY(n >= 0 && d_enc && d_pred && qp >= 0 && qp <= 51 && list >= 0 && list <= 3);
    if (!n) return X264GPU_OK;
    Q4 q = make_q4(qp, list);
    hipLaunchKernelGGL(k_dctq4x4, dim3((n + 63) / 64), dim3(256), 0, (hipStream_t)stream, d_enc, d_pred, n, q, d_coef, d_levels, d_recon);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_mc_luma(const uint8_t *d_planes00, size_t plane_bytes, int stride, const int32_t *d_xy,
                    const int32_t *d_mv, int n, int w, int h, uint8_t *d_out, void *stream)
{
    ARG_TRY(n >= 0 && d_planes00 && d_xy && d_mv && d_out && (w == 4 || w == 8 || w == 16) && (h == 4 || h == 8 || h == 16));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_mc_luma, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_planes00, plane_bytes, stride, d_xy, d_mv, n, w, h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_mc_avg(const uint8_t *d_a, const uint8_t *d_b, size_t bytes, int weight1, uint8_t *d_out, void *stream)
{
    ARG_TRY(d_a && d_b && d_out && !(bytes & 3) && weight1 >= -64 && weight1 <= 128);
    if (!bytes) return X264GPU_OK;
    hipLaunchKernelGGL(k_mc_avg, dim3((unsigned)((bytes / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_a, (const uint32_t *)d_b, bytes / 4, weight1, (uint32_t *)d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_mc_weight(const uint8_t *d_src, size_t bytes, int scale, int denom, int offset, uint8_t *d_out, void *stream)
{
    ARG_TRY(d_src && d_out && !(bytes & 3) && denom >= 0 && denom <= 7 && scale >= 0 && scale <= 255 && offset >= -128 && offset <= 127);
    if (!bytes) return X264GPU_OK;
    hipLaunchKernelGGL(k_mc_weight, dim3((unsigned)((bytes / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_src, bytes / 4, scale, denom, offset, (uint32_t *)d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_mc_chroma(const uint8_t *d_nv12_00, int stride, const int32_t *d_xy, const int32_t *d_mv, int n,
                      int w, int h, uint8_t *d_out, void *stream)
{
    ARG_TRY(n >= 0 && d_nv12_00 && d_xy && d_mv && d_out && (w == 4 || w == 8) && (h == 4 || h == 8));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_mc_chroma, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_nv12_00, stride, d_xy, d_mv, n, w, h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

}  // extern "C"

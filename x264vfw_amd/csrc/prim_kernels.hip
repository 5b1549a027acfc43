// prim_kernels.hip — Tier-1 batch primitives of include/x264gpu.h ("checkasm" surface).
// Thin kernels around the wave-level device library (dsp.hip.h, mc.hip.h) that the frame pipeline uses,
// so each primitive is parity-tested against oracle/ with exactly the device code that ships.
#include "common.hip.h"
#include "mc.hip.h"
#include "k_mb.hip.h"          // the motion cache + intra helpers cabac_rd.hip.h builds on
#include "trellis.hip.h"
#include <math.h>
#include <mutex>
#include <vector>
#include <stddef.h>

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

// ---- trellis quantiser primitive (trellis.hip.h): blocks in, levels out, against caller-supplied context variables ----
}  // extern "C"

namespace x264gpu {

// x264_rdo_init's unary tables and x264_trellis_lambda2_tab, built once per DEVICE and kept there (GOP slots of one session live on
// several devices, host/encoder.cpp: every device's macroblock loop must read its own copy).  One allocation holds the three tables.
static int trellis_tables(TrellisTab *out)
{
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static TrellisTab tt[MAXDEV] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ARG_TRY(dev >= 0 && dev < MAXDEV);
    std::lock_guard<std::mutex> lock(mu);
    if (!tt[dev].size_unary) {
        static const uint16_t ent[128] = {
#include "cabac_entropy.inc"
        };
        static const uint8_t trans_lps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                               24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
        auto next = [&](int st, int b) { const int s = st >> 1, mps = st & 1; return (mps ^ b) ? (trans_lps[s] << 1) | (s == 0 ? mps ^ 1 : mps) : ((s < 62 ? s + 1 : 62) << 1) | mps; };
        struct Host { int l2[104]; int qt[52 * TRELLIS_QT_ROW]; uint16_t su[15 * 128]; uint8_t tu[15 * 128]; };       // (qt directly behind l2: trellis.hip.h finds it there)
        static Host hst;
        for (int prefix = 0; prefix < 15; prefix++)
            for (int c0 = 0; c0 < 128; c0++) {
                int bits = 0, ctx = c0;
                for (int i = 1; i < prefix; i++) { bits += ent[ctx ^ 1]; ctx = next(ctx, 1); }
                if (prefix > 0 && prefix < 14) { bits += ent[ctx]; ctx = next(ctx, 0); }
                hst.su[prefix * 128 + c0] = (uint16_t)(bits + 256); hst.tu[prefix * 128 + c0] = (uint8_t)ctx;
            }
        for (int qp = 0; qp < 52; qp++) { hst.l2[qp] = (int)(0.85 * 0.85 * pow(2.0, qp / 3.0 + 6.0) + 0.5); hst.l2[52 + qp] = (int)(0.65 * 0.65 * pow(2.0, qp / 3.0 + 6.0) + 0.5); }
        // quantiser, rounding offset, inverse and distortion weight of every coefficient class at every quantiser (trellis.hip.h used to divide for
        // them at the top of every call): classes 0..2 of 4x4 blocks, 3..8 of 8x8 blocks, 9 of DC blocks; { mf, bias, unq, w } each
        {
            static const int q4[6][3] = { { 13107, 8066, 5243 }, { 11916, 7490, 4660 }, { 10082, 6554, 4194 }, { 9362, 5825, 3647 }, { 8192, 5243, 3355 }, { 7282, 4559, 2893 } };
            static const int q8[6][6] = { { 13107, 11428, 20972, 12222, 16777, 15481 }, { 11916, 10826, 19174, 11058, 14980, 14290 }, { 10082, 8943, 15978, 9675, 12710, 11985 },
                                          { 9362, 8228, 14913, 8931, 11984, 11259 },    { 8192, 7346, 13159, 7740, 10486, 9777 },    { 7282, 6428, 11570, 6830, 9118, 8640 } };
            static const int w4[3] = { 800, 320, 128 }, w8[6] = { 256, 201, 656, 227, 410, 363 };
            auto srd = [](int x, int sh) { return sh <= 0 ? x << -sh : (x + (1 << (sh - 1))) >> sh; };
            for (int qp = 0; qp < 52; qp++) {
                int *row = hst.qt + qp * TRELLIS_QT_ROW;
                for (int cl = 0; cl < 3; cl++) {
                    const int mf = srd(q4[qp % 6][cl], qp / 6 - 1);
                    row[cl * 4 + 0] = mf; row[cl * 4 + 1] = (1 << 15) / mf; row[cl * 4 + 2] = (int)((1ull << (qp / 6 + 15 + 8)) / (unsigned long long)q4[qp % 6][cl]); row[cl * 4 + 3] = w4[cl];
                }
                for (int cl = 0; cl < 6; cl++) {
                    const int mf = srd(q8[qp % 6][cl], qp / 6);
                    int *e = row + (3 + cl) * 4;
                    e[0] = mf; e[1] = (1 << 15) / mf; e[2] = (int)((1ull << (qp / 6 + 16 + 8)) / (unsigned long long)q8[qp % 6][cl]); e[3] = w8[cl];
                }
                {
                    const int m0 = srd(q4[qp % 6][0], qp / 6 - 1);
                    int *e = row + 9 * 4;
                    e[0] = m0 >> 1; e[1] = ((1 << 15) / m0) << 1; e[2] = (int)((1ull << (qp / 6 + 15 + 8)) / (unsigned long long)q4[qp % 6][0]) << 1; e[3] = 256;
                }
            }
        }
        char *a = nullptr;
        HIP_TRY(hipMalloc((void **)&a, sizeof(Host)));
        const hipError_t ce = hipMemcpy(a, &hst, sizeof(Host), hipMemcpyHostToDevice);
        if (ce != hipSuccess) { (void)hipFree(a); HIP_TRY(ce); }
        tt[dev].lambda2 = (const int *)(a + offsetof(Host, l2)); tt[dev].size_unary = (const uint16_t *)(a + offsetof(Host, su)); tt[dev].trans_unary = (const uint8_t *)(a + offsetof(Host, tu));
    }
    *out = tt[dev];
    return X264GPU_OK;
}

// The CABAC chain table (cabac_rd.hip.h cab_chain): what k = 0..8 bins on ONE context variable leave behind — entry ((2^k - 1) + pattern) * 128 + variable
// holds the variable after the bins (first bin = bit 0 of the pattern) in bits 0..6 and their cost (1/256 bit) above.  It turns the serial bin-by-bin
// state machine of the size-only coder into one lookup per eight bins; 261 KB, built once per device, read through the scalar cache / L2.
int cabac_chain_table(const uint32_t **out)
{
    constexpr int MAXDEV = 64;
    static std::mutex mu;
    static const uint32_t *tab[MAXDEV] = {};
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    ARG_TRY(dev >= 0 && dev < MAXDEV);
    std::lock_guard<std::mutex> lock(mu);
    if (!tab[dev]) {
        static const uint16_t ent[128] = {
#include "cabac_entropy.inc"
        };
        static const uint8_t trans_lps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                               24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
        static std::vector<uint32_t> h;
        if (h.empty()) {
            h.assign((size_t)CAB_CHAIN_ENTRIES, 0u);
            for (int k = 0; k <= 8; k++)
                for (int pat = 0; pat < (1 << k); pat++)
                    for (int st0 = 0; st0 < 128; st0++) {
                        int st = st0, bits = 0;
                        for (int i = 0; i < k; i++) {
                            const int b = (pat >> i) & 1, sg = st >> 1, mps = st & 1;
                            if (sg > 63) break;
                            if (mps ^ b) { bits += ent[2 * sg + 1]; st = (trans_lps[sg] << 1) | (sg == 0 ? mps ^ 1 : mps); }
                            else { bits += ent[2 * sg]; st = ((sg < 62 ? sg + 1 : 62) << 1) | mps; }
                        }
                        h[((size_t)((1 << k) - 1) + pat) * 128 + st0] = (uint32_t)(st & 127) | ((uint32_t)bits << 7);
                    }
        }
        uint32_t *a = nullptr;
        HIP_TRY(hipMalloc((void **)&a, h.size() * sizeof(uint32_t)));
        const hipError_t ce = hipMemcpy(a, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
        if (ce != hipSuccess) { (void)hipFree(a); HIP_TRY(ce); }
        tab[dev] = a;
    }
    *out = tab[dev];
    return X264GPU_OK;
}

// the same tables for the macroblock loop (encoder.hip)
int trellis_table_ptrs(const uint16_t **su, const uint8_t **tu, const int **l2)
{
    TrellisTab tt;
    const int rc = trellis_tables(&tt);
    if (rc == X264GPU_OK) { *su = tt.size_unary; *tu = tt.trans_unary; *l2 = tt.lambda2; }
    return rc;
}

template <int CAT>
__global__ void __launch_bounds__(64) k_trellis_blocks(const int16_t *coefs, int nblk, int qp, int intra, const uint8_t *states, TrellisTab tt, int16_t *levels, uint8_t *nz)
{
    constexpr int NC = CAT == 5 ? 64 : CAT == 3 ? 4 : 16;
    __shared__ int16_t buf[8 * 64];
    __shared__ uint8_t st[460];
    const int lane = threadIdx.x;
    for (int i = lane; i < 460; i += 64) st[i] = states[i];
    const uint32_t model = cab_model(lane);
    __syncthreads();
    const int sig_off = CAT == 0 ? 105 : CAT == 1 ? 120 : CAT == 2 ? 134 : CAT == 3 ? 149 : CAT == 4 ? 152 : 402;
    const int last_off = CAT == 0 ? 166 : CAT == 1 ? 181 : CAT == 2 ? 195 : CAT == 3 ? 210 : CAT == 4 ? 213 : 417;
    const int abs_off = CAT == 0 ? 227 : CAT == 1 ? 237 : CAT == 2 ? 247 : CAT == 3 ? 257 : CAT == 4 ? 266 : 426;
    // the category's context variables in the role layout of cabac_rd.hip.h (cab_locate): this lane's byte of r / r8
    uint32_t reg = 0;
    for (int ctx = 0; ctx < 460; ctx++) {
        int rg, ln, sh;
        if (!cab_locate(ctx, rg, ln, sh) || ln != lane) continue;
        const bool mine = (ctx >= sig_off && ctx < sig_off + 16) || (ctx >= last_off && ctx < last_off + 16) || (ctx >= abs_off && ctx < abs_off + 10);
        if (mine && rg == (CAT == 5 ? 2 : 1)) reg |= (uint32_t)st[ctx] << sh;
    }
    for (int b0 = 0; b0 < nblk; b0 += 8) {
        const int nb = min(8, nblk - b0);
        for (int i = lane; i < nb * NC; i += 64) buf[i] = coefs[(size_t)b0 * NC + i];
        __syncthreads();
        const unsigned m = trellis_blocks<CAT>((lds_i16 *)buf, NC, nb, qp, intra != 0, model, tt, reg);
        __syncthreads();
        for (int i = lane; i < nb * NC; i += 64) levels[(size_t)b0 * NC + i] = buf[i];
        if (lane < nb) nz[b0 + lane] = (uint8_t)((m >> lane) & 1);
        __syncthreads();
    }
}

}  // namespace x264gpu

extern "C" {

int x264gpu_trellis_blocks(const int16_t *d_coefs, int nblk, int cat, int qp, int intra, const uint8_t *d_states460, int16_t *d_levels, uint8_t *d_nz, void *stream)
{
    ARG_TRY(d_coefs && d_states460 && d_levels && d_nz && nblk >= 0 && cat >= 0 && cat <= 5 && qp >= 0 && qp <= 51);
    if (!nblk) return X264GPU_OK;
    TrellisTab tt;
    const int rc = trellis_tables(&tt);
    if (rc != X264GPU_OK) return rc;
    const hipStream_t st = (hipStream_t)stream;
    switch (cat) {
    case 0: hipLaunchKernelGGL(k_trellis_blocks<0>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    case 1: hipLaunchKernelGGL(k_trellis_blocks<1>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    case 2: hipLaunchKernelGGL(k_trellis_blocks<2>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    case 3: hipLaunchKernelGGL(k_trellis_blocks<3>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    case 4: hipLaunchKernelGGL(k_trellis_blocks<4>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    default: hipLaunchKernelGGL(k_trellis_blocks<5>, dim3(1), dim3(64), 0, st, d_coefs, nblk, qp, intra, d_states460, tt, d_levels, d_nz); break;
    }
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}


/* the level walk of the CABAC pricing (cabac_rd.hip.h cab_levels_all) as a primitive: n macroblocks' levels (x264gpu_mb layout, 416 each), what to cover
 * per case (d_what[5 n]: luma category 2 / 5 / 1 / -1, luma block mask, chroma AC block mask, chroma DC plane mask, luma DC flag), the role-indexed
 * context registers r / r8 in and out (64 lanes each), the bits (1/256) out */
}  // extern "C"
namespace x264gpu {
__global__ void k_cab_level_walk(const int16_t *lv, const int *what, const uint32_t *r_in, const uint32_t *r8_in, uint32_t *r_out, uint32_t *r8_out, int *bits, const uint32_t *ctab)
{
    __shared__ int16_t l[X264GPU_MB_LEVELS];
    const int i = blockIdx.x, lane = threadIdx.x;
    for (int j = lane; j < X264GPU_MB_LEVELS; j += 64) l[j] = lv[(size_t)i * X264GPU_MB_LEVELS + j];
    __syncthreads();
    Cab cb; cb.a = 0; cb.r = r_in[i * 64 + lane]; cb.r8 = r8_in[i * 64 + lane]; cb.f8 = 0; cb.f8v = 0;
    const uint32_t model = cab_model(lane);
    CabLv W = { what[i * 5], (unsigned)what[i * 5 + 1], (unsigned)what[i * 5 + 2], (unsigned)what[i * 5 + 3], what[i * 5 + 4] != 0 };
    Prof pf;
    cab_levels_all(cb, model, lane, l, W, ctab, pf);
    r_out[i * 64 + lane] = cb.r; r8_out[i * 64 + lane] = cb.r8;
    const int t = cab_total(cb);
    if (lane == 0) bits[i] = t;
}
}  // namespace x264gpu
extern "C" {
int x264gpu_cabac_level_walk(const int16_t *d_levels, const int32_t *d_what, int n, const uint32_t *d_r, const uint32_t *d_r8, uint32_t *d_r_out, uint32_t *d_r8_out, int32_t *d_bits, void *stream)
{
    ARG_TRY(d_levels && d_what && d_r && d_r8 && d_r_out && d_r8_out && d_bits && n >= 0);
    if (!n) return X264GPU_OK;
    const uint32_t *ctab = nullptr;
    { const int rc = x264gpu::cabac_chain_table(&ctab); if (rc != X264GPU_OK) return rc; }
    hipLaunchKernelGGL(x264gpu::k_cab_level_walk, dim3(n), dim3(64), 0, (hipStream_t)stream, d_levels, d_what, d_r, d_r8, d_r_out, d_r8_out, d_bits, ctab);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}
}  // extern "C"

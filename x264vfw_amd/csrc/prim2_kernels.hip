// prim2_kernels.hip — Tier-1 primitives, part 2: the 8x8 transform path (A6-A8: sub8x8_dct8, quant_8x8,
// dequant_8x8, add8x8_idct8) and the intra predictors (A5) as batch kernels.
//
// 8x8 layout: lane = (block, row) — 8 lanes own one 8x8 block, each lane holds one row of 8 samples in
// registers.  Row transforms are in-lane 8-point butterflies; column transforms run after an 8x8 transpose
// across the 8 lanes (three exchange stages: DPP quad_perm for lane^1 / lane^2, a cross-lane shuffle for
// lane^4).  Quantiser tables have six position classes (normAdjust8x8); each lane keeps the four it needs.
#include "common.hip.h"
#include "intra.hip.h"
#include "dsp8.hip.h"

using namespace x264gpu;

namespace {

__global__ __launch_bounds__(256) void k_dctq8x8(const uint8_t *__restrict__ enc, const uint8_t *__restrict__ pred, int n, Q8 q,
                                                 int16_t *__restrict__ coef, int16_t *__restrict__ levels, uint8_t *__restrict__ recon)
{
    const int lane = threadIdx.x & 63;
    const int blk = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (lane >> 3), row = lane & 7;
    const bool valid = blk < n;
    uint2 pe = make_uint2(0, 0), pp = make_uint2(0, 0);
    if (valid) { pe = *(const uint2 *)(enc + (size_t)blk * 64 + row * 8); pp = *(const uint2 *)(pred + (size_t)blk * 64 + row * 8); }
    int e[8], p[8], v[8];
    unpack8(pe.x, pe.y, e); unpack8(pp.x, pp.y, p);
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
    fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);       // natural layout: lane = row, reg = col
    if (coef && valid)
#pragma unroll
        for (int i = 0; i < 8; i++) coef[(size_t)blk * 64 + row * 8 + i] = (int16_t)v[i];
    int mf[4], bs[4], dq[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { const int k = class8(row, c); mf[c] = q.mf[k]; bs[c] = q.bias[k]; dq[c] = q.dq[k]; }
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
    if (levels && valid)
#pragma unroll
        for (int i = 0; i < 8; i++) levels[(size_t)blk * 64 + row * 8 + i] = (int16_t)v[i];
    const int qb = q.qp / 6 - 6;
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
    inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);       // 8.5.13: rows first, then columns
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
    if (recon && valid) {
        uint2 o;
        o.x = pack4_clip8lo(v); o.y = pack4_clip8hi(v);
        *(uint2 *)(recon + (size_t)blk * 64 + row * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// intra predictors: one wavefront per block; neighbours staged in per-wave LDS exactly as in k_intra
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_intra_predict(int kind, const uint8_t *__restrict__ plane, int stride,
                                                       const int32_t *__restrict__ xy, const int32_t *__restrict__ mode,
                                                       const int32_t *__restrict__ avail, int n, uint8_t *__restrict__ out)
{
    __shared__ uint8_t s_nb[4][NB_SIZE];
    __shared__ __attribute__((aligned(4))) uint8_t s_tile[4][5 * 16];
    __shared__ uint8_t s_u[4][U_SIZE];
    __shared__ uint8_t s_cnb[4][CNB_SIZE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;
    if (idx >= n) return;
    const uint8_t *blk = plane + (size_t)xy[2 * idx + 1] * stride + xy[2 * idx];
    const int m = mode[idx];
    if (kind == 0) {
        uint8_t *nb = s_nb[wave];
        if (lane < 17) nb[NB_TOP - 1 + lane] = blk[-(long)stride - 1 + lane];
        else if (lane >= 32 && lane < 48) nb[NB_LEFT + lane - 32] = blk[(long)(lane - 32) * stride - 1];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const Pred16 pp = pred16_setup(nb, lane);
        const int x0 = (lane & 3) * 4, y = lane >> 2;
        *(uint32_t *)(out + (size_t)idx * 256 + y * 16 + x0) = pred16_row4(nb, pp, m, x0, y);
    } else if (kind == 1) {
        uint8_t *nb = s_cnb[wave];
        if (lane < 9) nb[CNB_TOP - 1 + lane] = blk[-(long)stride - 1 + lane];
        else if (lane >= 16 && lane < 24) nb[CNB_LEFT + lane - 16] = blk[(long)(lane - 16) * stride - 1];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const PredC pc = predc_setup(nb);
        if (lane < 16) {
            const int i = lane >> 2, j = lane & 3;
            *(uint32_t *)(out + (size_t)idx * 64 + ((i >> 1) * 4 + j) * 8 + (i & 1) * 4) = predc_row4(nb, pc, m, i, j);
        }
    } else {
        // 4x4: tile rows -1..3, columns -1..7 at offset (row+1)*16 + 4 + col
        uint8_t *tile = s_tile[wave] + 16 + 4;
        const int av = avail[idx];
        if (lane < 9) tile[-16 - 1 + lane] = blk[-(long)stride - 1 + lane];
        else if (lane >= 16 && lane < 20) tile[(lane - 16) * 16 - 1] = blk[(long)(lane - 16) * stride - 1];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        pred4_build_u(s_u[wave], tile, 16, av, lane);
        // DC variants (modes 9..11 of the oracle) are selected by availability inside pred4_build_u: mode 2
        const int mm = m > 8 ? 2 : m;
        if (lane < 4) {
            const uint32_t t4 = ((const uint32_t *)c_pred4_table.t)[mm * 4 + lane];
            *(uint32_t *)(out + (size_t)idx * 16 + lane * 4) = pred4_row4(s_u[wave], t4);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// pixel_hadamard_ac ([x264-upstream] common/pixel.c; psy-RD's energy term): AC energy of the 4x4 and of the 8x8
// Hadamard transforms of a SOURCE block.  R8 layout: lane = (8x8 sub-block, row), 8 pixels per lane; the packed-16 SATD / SA8D helpers
// run on the pixels themselves (prediction = 0), both DC terms are the sub-block's pixel sum.  One wave per block (w x h in
// {16x16, 8x16, 16x8, 8x8}); out = ((sum8 >> 2) << 32) | (sum4 >> 1) as x264 packs it.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_hadamard_ac(const uint8_t *__restrict__ a, int n, int w, int h, uint64_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63, idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= n) return;
    const int nsub = (w >> 3) * (h >> 3), sub = lane >> 3, row = lane & 7;
    const bool act = sub < nsub;
    const int sx = act ? (sub % (w >> 3)) * 8 : 0, sy = act ? (sub / (w >> 3)) * 8 : 0;
    const uint2 v = act ? *(const uint2 *)(a + (size_t)idx * w * h + (size_t)(sy + row) * w + sx) : make_uint2(0, 0);
    int p[8];
    unpack8(v.x, v.y, p);
    const int psum = p[0] + p[1] + p[2] + p[3] + p[4] + p[5] + p[6] + p[7];
    const int h4 = satd4_half(v.x, 0u, lane) + satd4_half(v.y, 0u, lane), h8 = sa8d_r8_half(v.x, v.y, 0u, 0u, lane);
    // per wave: sum over the active sub-blocks of (2 * shares - pixel sum)
    const unsigned s4 = (unsigned)wave_sum(act ? 2 * h4 - psum : 0), s8 = (unsigned)wave_sum(act ? 2 * h8 - psum : 0);
    if (lane == 0) out[idx] = ((uint64_t)(s8 >> 2) << 32) | (s4 >> 1);
}
}  // namespace

extern "C" {

int x264gpu_dctq8x8(const uint8_t *d_enc, const uint8_t *d_pred, int n, int qp, int list, int16_t *d_coef, int16_t *d_levels,
                    uint8_t *d_recon, void *stream)
{
    ARG_TRY(n >= 0 && d_enc && d_pred && qp >= 0 && qp <= 51 && (list == 0 || list == 1));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_dctq8x8, dim3((n + 31) / 32), dim3(256), 0, (hipStream_t)stream, d_enc, d_pred, n, make_q8(qp, list), d_coef, d_levels, d_recon);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_intra_predict(int kind, const uint8_t *d_plane, int stride, const int32_t *d_xy, const int32_t *d_mode,
                          const int32_t *d_avail, int n, uint8_t *d_out, void *stream)
{
    ARG_TRY(kind >= 0 && kind <= 2 && d_plane && d_xy && d_mode && d_out && n >= 0 && (kind != 2 || d_avail));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_intra_predict, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, kind, d_plane, stride, d_xy, d_mode, d_avail, n, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_pixel_hadamard_ac(const uint8_t *d_a, int n, int w, int h, uint64_t *d_out, void *stream)
{
    ARG_TRY(n >= 0 && d_a && d_out && (w == 8 || w == 16) && (h == 8 || h == 16));
    if (!n) return X264GPU_OK;
    hipLaunchKernelGGL(k_hadamard_ac, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, d_a, n, w, h, d_out);
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

}  // extern "C"

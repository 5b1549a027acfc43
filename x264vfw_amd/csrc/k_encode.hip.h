// k_encode.hip.h — frame-level kernels around the macroblock loop (k_mb.hip.h): ingest (A1), per-macroblock quantisers (AQ, caller
// offsets), QP_Y inheritance.  Restates oracle/encoder.c ingest / compute_mb_qp / settle_mb_qp bit-exactly.
#pragma once
#include "enc_common.hip.h"

namespace x264gpu {

// ------------------------------------------------------------------------------------------------
// stage 0: ingest I420 -> MB-aligned luma + NV12 chroma with edge replication (A1)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ingest(EncK k)
{
    const int s = blockIdx.z;
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= k.cw) return;
    const uint8_t *src = k.i420 + (size_t)s * ((size_t)k.w * k.h * 3 / 2);
    const uint8_t *sy = src + (size_t)min(y, k.h - 1) * k.w;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = sy[min(x + i, k.w - 1)];
    *(uint32_t *)(k.fenc_y + (size_t)s * k.fency_bytes + (size_t)y * k.fs + x) = pack4(v);
    if (y < k.ch / 2) {
        // 4 output bytes = 2 chroma samples (U,V interleaved)
        const int cwid = k.w / 2, chgt = k.h / 2;
        const uint8_t *su = src + (size_t)k.w * k.h + (size_t)min(y, chgt - 1) * cwid, *sv = su + (size_t)cwid * chgt;
        int c0 = min(x / 2, cwid - 1), c1 = min(x / 2 + 1, cwid - 1);
        int u[4] = { su[c0], sv[c0], su[c1], sv[c1] };
        *(uint32_t *)(k.fenc_uv + (size_t)s * k.fencuv_bytes + (size_t)y * k.fs + x) = pack4(u);
    }
}

// ------------------------------------------------------------------------------------------------
// Adaptive quantisation, mode 1 (oracle compute_mb_qp; x264_adaptive_quant_frame): one 16-lane row per macroblock — lane r sums row
// r of the luma macroblock and, for r < 8, row r of both chroma planes; the AC energy -> x264_log2 -> quantiser offset, in x264's single floats.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_aq(EncK k)
{
    const int lane = threadIdx.x & 63, r = lane & 15, s = blockIdx.y;
    const int mbi = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool valid = mbi < k.nmb;
    const int mi = valid ? mbi : 0, mbx = mi % k.mbw, mby = mi / k.mbw;
    const uint8_t *y = k.fenc_y + (size_t)s * k.fency_bytes + (size_t)(mby * 16 + r) * k.fs + mbx * 16;
    unsigned sum = 0, sqr = 0, su = 0, squ = 0, sv = 0, sqv = 0;
    const uint4 v = *(const uint4 *)y;
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int b = 0; b < 4; b++) { const unsigned p = (w[i] >> (8 * b)) & 0xff; sum += p; sqr += p * p; }
    if (r < 8) {
        const uint8_t *uv = k.fenc_uv + (size_t)s * k.fencuv_bytes + (size_t)(mby * 8 + r) * k.fs + mbx * 16;
        const uint4 c = *(const uint4 *)uv;
        const uint32_t cw[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int b = 0; b < 4; b++) { const unsigned p = (cw[i] >> (8 * b)) & 0xff; if (b & 1) { sv += p; sqv += p * p; } else { su += p; squ += p * p; } }
    }
    sum = (unsigned)row16_sum((int)sum); sqr = (unsigned)row16_sum((int)sqr);
    su = (unsigned)row16_sum((int)su); squ = (unsigned)row16_sum((int)squ); sv = (unsigned)row16_sum((int)sv); sqv = (unsigned)row16_sum((int)sqv);
    const unsigned energy = (sqr - (sum * sum >> 8)) + (squ - (su * su >> 6)) + (sqv - (sv * sv >> 6));
    const float adj = f_mul(k.aq_strength, f_sub(x264_log2(energy ? energy : 1u), 14.427f));
    if (valid && r == 0) k.mbqp[(size_t)s * k.nmb + mbi] = (uint8_t)x264_mb_qp(slice_qpm(k, s), adj);          // x264_ratecontrol_mb_qp: (int)(qpm + offset + 0.5f)
}

// quantiser offsets decided by the lookahead (AQ - macroblock-tree, single floats) -> per-macroblock quantisers (oracle compute_mb_qp, ext_off)
__global__ __launch_bounds__(256) void k_apply_qp_offsets(EncK k, const float *__restrict__ off)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i < k.nmb) k.mbqp[(size_t)s * k.nmb + i] = (uint8_t)(off ? x264_mb_qp(slice_qpm(k, s), off[(size_t)s * k.nmb + i]) : slice_qp(k, s));   // no offsets: the slice's (or the stream's) quantiser as it is
}

// QP_Y inheritance (oracle settle_mb_qp, 7.4.5): a macroblock that sends no mb_qp_delta takes its predecessor's quantiser; one wave per
// stream walks the records 64 at a time (ballot of the macroblocks that keep their own value, highest one at or below each lane).
__global__ __launch_bounds__(64) void k_settle_qp(EncK k)
{
    // every macroblock is a function of the previous one's settled qp: "mine" (it codes a delta), "the previous" (nothing coded), or
    // "min(mine, previous)" (an I16x16 with nothing coded, DC included, never RAISES the quantiser: x264's qp_delta writers).  Those
    // compose — (is_const, v) with previous == min(255, .) — so a wave scan settles 64 macroblocks per step
    const int lane = threadIdx.x, s = blockIdx.x;
    // one wave per (stream, slice): the chain starts from the slice quantiser at every slice
    const int nsl = k.slices > 1 ? k.slices : 1, mb0 = ((k.mbh * (int)blockIdx.y + nsl / 2) / nsl) * k.mbw, mb1 = ((k.mbh * ((int)blockIdx.y + 1) + nsl / 2) / nsl) * k.mbw;
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    int carry = k.mbqp[(size_t)s * k.nmb + mb0];          // the slice quantiser = the first macroblock's (x264_slice_write)
    for (int base = mb0; base < mb1; base += 64) {
        const int i = base + lane;
        const bool in = i < mb1;
        int v = 255;
        bool cst = false;
        if (in) {
            const x264gpu_mb *m = mbs + i;
            const bool coded = m->cbp_luma || m->cbp_chroma;
            if (m->type == X264GPU_MB_I16x16) { v = m->qp; cst = coded || ((m->nnz >> 24) & 1); }
            else if (coded) { v = m->qp; cst = true; }
        }
        for (int d = 1; d < 64; d <<= 1) {
            const int pv = __shfl_up(v, d);
            const int pc = __shfl_up((int)cst, d);
            if (lane >= d && !cst) { v = min(v, pv); cst = pc != 0; }
        }
        const int settled = cst ? v : min(v, carry);
        if (in) mbs[i].qp = (uint8_t)settled;
        carry = __shfl(settled, min(mb1 - base, 64) - 1);
    }
}

}  // namespace x264gpu

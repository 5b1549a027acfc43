// k_encode.cuh — macroblock encode kernels: motion compensation + 4x4 integer DCT + deadzone quant +
// decimation + dequant + iDCT + reconstruction (A6/A7/A8/A10) for inter macroblocks, and the
// reconstructed-neighbour intra stage (A5) run as a 2-D wavefront by one workgroup per stream.
// Restates oracle/encoder.c encode_inter_mb / intra_mb bit-exactly.
#pragma once
#include "enc_common.cuh"

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
// stage 2: inter macroblock encode, one wavefront per macroblock (Z layout)
// ------------------------------------------------------------------------------------------------
// AQ = per-macroblock quantisers (k.mbqp set): its own instantiation, so that constant-quantiser sessions keep reading the slice's
// quantiser tables straight from the kernel arguments instead of through a run-time select of two structures
template <bool AQ>
__global__ __launch_bounds__(256) void k_encode_inter(EncK k)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;       // scalar: see k_analyse_p
    const int mbi = blockIdx.x * 4 + wave, s = blockIdx.y;
    if (mbi >= k.nmb) return;
    x264gpu_mb *mbp = k.mb + (size_t)s * k.nmb + mbi;
    if (mbp->type != X264GPU_MB_P_L0 && mbp->type != X264GPU_MB_P_8x8) return;       // wave-uniform
    const int mbx = mbi % k.mbw, mby = mbi / k.mbw, px = mbx * 16, py = mby * 16;
    const int mvx = mbp->mv[lane >> 4][0], mvy = mbp->mv[lane >> 4][1];     // luma: lane>>4 = 8x8 block (partition motion)
    int16_t *lv = k.levels + ((size_t)s * k.nmb + mbi) * X264GPU_MB_LEVELS;
    const uint8_t *fenc = k.fenc_y + (size_t)s * k.fency_bytes + (size_t)py * k.fs + px;
    const int j = lane & 3, blk = lane >> 2, zx = z_x0(lane), zy = z_y(lane);
    // quantiser of this macroblock: the slice's, or its own under AQ (wave-uniform table reads)
    const int mqp = AQ ? (int)mbp->qp : k.qp, mqpc = AQ ? (int)d_chroma_qp_table[min(max(mqp + k.chroma_qp_offset, 0), 51)] : k.qpc;
    const Q4 &q_luma_inter = AQ ? k.q4tab[mqp * 4 + 1] : k.q_luma_inter, &q_chroma_inter = AQ ? k.q4tab[mqpc * 4 + 3] : k.q_chroma_inter;
    const Q8 &q8_inter = AQ ? k.q8tab[mqp * 2 + 1] : k.q8_inter;

    // ---- luma ----
    const int refidx = mbp->ref[lane >> 4];         // the reference is per 8x8 block (mixed refs)
    const uint32_t pred = mc_luma_row4(ref_plane00(k, s, refidx), k.plane_bytes, k.rs, px + zx, py + zy, mvx, mvy);
    const uint32_t enc = *(const uint32_t *)(fenc + (size_t)zy * k.fs + zx);
    unsigned nnz = 0;
    int cbp_luma = 0;
    // transform size ([x264-upstream] analyse.c x264_mb_analyse_transform): SA8D vs SATD of the prediction error
    bool t8 = false;
    uint32_t elo = 0, ehi = 0, plo = 0, phi = 0;
    if (k.dct8x8) {
        z_to_r8(enc, lane, elo, ehi); z_to_r8(pred, lane, plo, phi);
        const int h8 = sa8d_r8_half(elo, ehi, plo, phi, lane);
        const int cost8 = (2 * wave_sum(lane < 32 ? h8 : 0) + 2) >> 2, cost4 = wave_sum(satd4_half(enc, pred, lane));
        t8 = cost8 < cost4;
    }
    if (t8) {
        // ---- 8x8 transform, R8 layout: lane = (8x8 block, row) on lanes 0..31 (upper half mirrors) ----
        const int row = lane & 7, i8 = (lane >> 3) & 3;
        int e[8], p[8], v[8];
        unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
        fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
        int mf[4], bs[4], dq[4];
        q8_row(q8_inter, row, mf, bs, dq);
        unsigned mlo = 0, mhi = 0, big = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
            const int z = c_zigzag8_inv[row * 8 + i];
            if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
            big |= abs(v[i]) > 1 ? 1u : 0u;
        }
        mlo = group8_or(mlo); mhi = group8_or(mhi); big = group8_or(big);
        const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
        bool keep = mask != 0;
        if (k.dct_decimate) {
            const int sc = keep ? (big ? 9 : decimate64_from_mask(mask)) : 0;
            const int mbscore = __builtin_amdgcn_readlane(sc, 0) + __builtin_amdgcn_readlane(sc, 8) + __builtin_amdgcn_readlane(sc, 16) + __builtin_amdgcn_readlane(sc, 24);
            keep = keep && sc >= 4 && mbscore >= 6;
        }
        // levels leave in the CAVLC-interleaved 4x4 form: scan index z -> block 4*i8 + (z & 3), position z >> 2
        if (lane < 32) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int z = c_zigzag8_inv[row * 8 + i];
                lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)(keep ? v[i] : 0);
            }
        }
        unsigned n4 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) n4 |= (mask & (0x1111111111111111ull << q)) ? 1u << q : 0u;
        if (!keep) n4 = 0;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const unsigned ng = (unsigned)__builtin_amdgcn_readlane((int)n4, g * 8);
            nnz |= ng << (4 * g);
            cbp_luma |= ng ? 1 << g : 0;
        }
        const int qb = q8_inter.qp / 6 - 6;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = keep ? dequant_one(v[i], dq[i & 3], qb) : 0;
        inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);        // 8.5.13: rows first, then columns
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
        if (lane < 32) {
            uint2 o;
            o.x = pack4_clip8lo(v); o.y = pack4_clip8hi(v);
            *(uint2 *)(rec_plane00(k, s) + (size_t)(py + (i8 >> 1) * 8 + row) * k.rs + px + (i8 & 1) * 8) = o;
        }
    } else {
    int e[4], p[4], v[4];
    unpack4(enc, e); unpack4(pred, p);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = e[i] - p[i];
    dct4_quad(v, lane);
    quant4_row(v, q_luma_inter, j);
    const unsigned mask = (unsigned)quad_or((int)scan_mask(v, j));
    const bool nz = mask != 0;
    bool keep = nz;
    if (k.dct_decimate) {
        const int big = quad_or(any_big(v) ? 1 : 0);
        const int sc = nz ? (big ? 9 : decimate_from_mask(mask, 0)) : 0;
        const int score8 = row16_sum(j == 0 ? sc : 0);                  // per 8x8 (= DPP row)
        bool any8 = row16_or(nz ? 1 : 0) != 0;
        const int mbscore = wave_sum(((lane & 15) == 0 && any8) ? score8 : 0);   // every coded 8x8 counts, kept or not
        if (any8 && score8 < 4) any8 = false;
        keep = nz && any8 && mbscore >= 6;
    }
    { int z[4] = { 0, 0, 0, 0 }; store_levels_scan(lv + blk * 16, keep ? v : z, j); }
    if (!keep) v[0] = v[1] = v[2] = v[3] = 0;
    dequant4_row(v, q_luma_inter, j);
    idct4_quad(v, lane);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] += p[i];
    *(uint32_t *)(rec_plane00(k, s) + (size_t)(py + zy) * k.rs + px + zx) = pack4_clip(v);
    const unsigned long long bal = __ballot(keep && j == 0);
#pragma unroll
    for (int b = 0; b < 16; b++) nnz |= (unsigned)((bal >> (4 * b)) & 1) << b;
#pragma unroll
    for (int i8 = 0; i8 < 4; i8++) cbp_luma |= ((nnz >> (4 * i8)) & 15) ? 1 << i8 : 0;
    }

    // ---- chroma (lanes 0..31: plane = lane>>4, block = (lane>>2)&3) ----
    const int c = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
    uint32_t pu, pv;
    mc_chroma_row4(ref_chroma00(k, s, mbp->ref[ci]), k.rs, mbx * 8 + cx0, mby * 8 + cyy, mbp->mv[ci][0], mbp->mv[ci][1], pu, pv);   // chroma 4x4 block ci <-> luma 8x8 ci
    const uint8_t *fuv = k.fenc_uv + (size_t)s * k.fencuv_bytes + (size_t)(mby * 8 + cyy) * k.fs + px + 2 * cx0;
    const uint2 fe = *(const uint2 *)fuv;
    const uint32_t cenc = nv12_pick(fe.x, fe.y, c), cpred = c ? pv : pu;
    int cbp_chroma = 0;
    const uint32_t crec = chroma_residual(cenc, cpred, q_chroma_inter, true, k.dct_decimate != 0, lane, lv, nnz, cbp_chroma);
    // interleave U (lanes 0..15) with V (lanes 16..31) and store 8 NV12 bytes from the U lanes
    const uint32_t other = (uint32_t)__shfl_xor((int)crec, 16);
    if (lane < 16) {
        const uint32_t u = crec, w = other;
        uint2 o;
        o.x = (u & 0xff) | ((w & 0xff) << 8) | ((u & 0xff00) << 8) | ((w & 0xff00) << 16);
        o.y = ((u >> 16) & 0xff) | (((w >> 16) & 0xff) << 8) | ((u >> 24) << 16) | ((w >> 24) << 24);
        *(uint2 *)(rec_chroma00(k, s) + (size_t)(mby * 8 + cyy) * k.rs + px + 2 * cx0) = o;
    }
    if (lane >= 32 && lane < 40) lv[X264GPU_LV_LUMA_DC + (lane - 32) * 2] = 0, lv[X264GPU_LV_LUMA_DC + (lane - 32) * 2 + 1] = 0;
    if (lane >= 40 && lane < 44) lv[408 + (lane - 40) * 2] = 0, lv[408 + (lane - 40) * 2 + 1] = 0;
    if (lane == 0) {
        mbp->nnz = nnz;
        mbp->cbp_luma = (uint8_t)cbp_luma;
        mbp->transform8x8 = (uint8_t)(t8 && cbp_luma);     // not transmitted without luma coefficients
        mbp->cbp_chroma = (uint8_t)cbp_chroma;
    }
}

// ------------------------------------------------------------------------------------------------
// Adaptive quantisation, mode 1 (oracle compute_mb_qp; x264_adaptive_quant_frame): one 16-lane row per macroblock — lane r sums row
// r of the luma macroblock and, for r < 8, row r of both chroma planes; the AC energy -> log2 in Q8 (table) -> quantiser offset.
// ------------------------------------------------------------------------------------------------
static __constant__ uint8_t c_aq_log2_lut[128] = {
#include "x264gpu_aq_lut.inc"
};
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
    const unsigned e1 = energy ? energy : 1u;
    const int lz = 31 - __builtin_clz(e1), lg = lz * 256 + c_aq_log2_lut[((e1 << (31 - lz)) >> 24) & 0x7f];
    const int adj = (k.aq_strength_q8 * (lg - 3693)) >> 8;
    if (valid && r == 0) k.mbqp[(size_t)s * k.nmb + mbi] = (uint8_t)min(max(slice_qp(k, s) + ((adj + 128) >> 8), 1), 51);
}

// quantiser offsets decided by the lookahead (AQ - macroblock-tree, Q8) -> per-macroblock quantisers (oracle compute_mb_qp, ext_off_q8)
__global__ __launch_bounds__(256) void k_apply_qp_offsets(EncK k, const int16_t *__restrict__ off)
{
    const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (i < k.nmb) k.mbqp[(size_t)s * k.nmb + i] = (uint8_t)min(max(slice_qp(k, s) + (off ? ((int)off[(size_t)s * k.nmb + i] + 128) >> 8 : 0), 1), 51);   // no offsets: per-stream quantisers only
}

// QP_Y inheritance (oracle settle_mb_qp, 7.4.5): a macroblock that sends no mb_qp_delta takes its predecessor's quantiser; one wave per
// stream walks the records 64 at a time (ballot of the macroblocks that keep their own value, highest one at or below each lane).
__global__ __launch_bounds__(64) void k_settle_qp(EncK k)
{
    const int lane = threadIdx.x, s = blockIdx.x;
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    int carry = slice_qp(k, s);
    for (int base = 0; base < k.nmb; base += 64) {
        const int i = base + lane;
        const bool in = i < k.nmb;
        int qp = 0;
        bool own = false;
        if (in) { const x264gpu_mb *m = mbs + i; qp = m->qp; own = m->type == X264GPU_MB_I16x16 || m->cbp_luma || m->cbp_chroma; }
        const unsigned long long owners = __ballot(own);
        const unsigned long long below = owners & (lane == 63 ? ~0ull : ((2ull << lane) - 1));
        const int src = below ? 63 - __builtin_clzll(below) : -1;
        const int from = __shfl(qp, src < 0 ? 0 : src);
        const int settled = src < 0 ? carry : from;
        if (in && !own) mbs[i].qp = (uint8_t)settled;
        const int last = min(k.nmb - base, 64) - 1;
        carry = __shfl(settled, last);
    }
}

}  // namespace x264gpu

// k_deblock.hip.h — in-loop deblocking filter (A9; H.264 8.7) as a 2-D wavefront.  The normative
// macroblock order (left, top and top-right neighbours first) makes this a d = x + 2y wavefront: one
// workgroup per stream (or several, few streams in flight), rows handed off through progress counters.  Each
// wavefront stages its macroblocks (+4 rows / 4 columns of neighbours) in LDS, runs the four vertical
// then four horizontal edges there (luma on lanes 0..15, chroma on lanes 16..31 in the same step), and
// writes back only the samples the standard lets this macroblock modify.
// Restates oracle/encoder.c deblock_frame + oracle/deblock.c bit-exactly.
#pragma once
#include "enc_common.hip.h"
#include "deblock_line.hip.h"

namespace x264gpu {

constexpr int DL_STRIDE = 24;                  // luma tile: rows -4..15, cols -4..15
constexpr int DL_ORG = 4 * DL_STRIDE + 4;
constexpr int DL_SIZE = 20 * DL_STRIDE;
constexpr int DC_STRIDE = 24;                  // chroma NV12 tile: rows -2..7, bytes -4..15
constexpr int DC_ORG = 2 * DC_STRIDE + 4;
constexpr int DC_SIZE = 10 * DC_STRIDE;
constexpr int WF_MAX_ROWS = 160;               // macroblock rows supported by the wavefront kernels (2304/16 = 144)

__device__ __forceinline__ bool mb_is_intra(int type) { return type == X264GPU_MB_I4x4 || type == X264GPU_MB_I8x8 || type == X264GPU_MB_I16x16; }

// boundary strength between 4x4 block (pbx,pby) of P and (qbx,qby) of Q (oracle edge_bs)
// k.refpic: P pictures under --weightp may hold a picture at several list-0 indices — the filter compares reference PICTURES
__device__ __forceinline__ int edge_bs(const x264gpu_mb *P, int pbx, int pby, const x264gpu_mb *Q, int qbx, int qby, bool mb_edge, const EncK &k)
{
    if (mb_is_intra(P->type) || mb_is_intra(Q->type)) return mb_edge ? 4 : 3;
    // transform8x8: "the 8x8 block containing the sample" has coefficients (8.7.2.1) = its cbp_luma bit
    const int pn = P->transform8x8 ? (P->cbp_luma >> ((pby >> 1) * 2 + (pbx >> 1))) & 1
                                   : (P->nnz >> (((pby >> 1) * 2 + (pbx >> 1)) * 4 + (pby & 1) * 2 + (pbx & 1))) & 1;
    const int qn = Q->transform8x8 ? (Q->cbp_luma >> ((qby >> 1) * 2 + (qbx >> 1))) & 1
                                   : (Q->nnz >> (((qby >> 1) * 2 + (qbx >> 1)) * 4 + (qby & 1) * 2 + (qbx & 1))) & 1;
    if (pn || qn) return 2;
    const int pi = (pby >> 1) * 2 + (pbx >> 1), qi = (qby >> 1) * 2 + (qbx >> 1);
    {
        int rp = P->ref[pi], rq = Q->ref[qi];
        if (k.wp_any) { rp = rp >= 0 ? k.refpic[rp & 7] : rp; rq = rq >= 0 ? k.refpic[rq & 7] : rq; }
        if (rp != rq) return 1;
    }
    if (abs(P->mv[pi][0] - Q->mv[qi][0]) >= 4 || abs(P->mv[pi][1] - Q->mv[qi][1]) >= 4) return 1;
    // B slices (x264 deblock_strength_c with bframe): list 1 index by index as well; the lists never share a picture
    if (P->type >= X264GPU_MB_B_DIRECT && Q->type >= X264GPU_MB_B_DIRECT) {
        if (P->ref1[pi] != Q->ref1[qi]) return 1;
        const int px = P->ref1[pi] < 0 ? 0 : P->mv1[pi][0], py = P->ref1[pi] < 0 ? 0 : P->mv1[pi][1];
        const int qx = Q->ref1[qi] < 0 ? 0 : Q->mv1[qi][0], qy = Q->ref1[qi] < 0 ? 0 : Q->mv1[qi][1];
        if (abs(px - qx) >= 4 || abs(py - qy) >= 4) return 1;
    }
    return 0;
}

__device__ __forceinline__ void filter_chroma_line(uint8_t *pix, int xs, int alpha, int beta, int tc0, int bs)
{
    const int p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) return;
    if (bs < 4) {
        const int tc = tc0 + 1;
        const int delta = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
        pix[-xs] = (uint8_t)clip_u8(p0 + delta);
        pix[0] = (uint8_t)clip_u8(q0 - delta);
    } else {
        pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}

// ------------------------------------------------------------------------------------------------
// Two macroblocks per wavefront: lanes 0..31 filter macroblock (x, row r0), lanes 32..63 macroblock (x - 2, row r0 + 1) — the
// row below runs two macroblocks behind, which is exactly the top-right dependency, so a wave walks a pair of rows in lock-step
// and the per-macroblock chain of LDS round trips (the kernel is latency-bound: 1.3 G VALU instructions per 256 frames but
// 12 ms) serves twice the pixels.  Everything that was wave-uniform per macroblock is per half here.
// ------------------------------------------------------------------------------------------------
struct Deblock2Lds {
    uint8_t lt[16][2][DL_SIZE];
    uint8_t ct[16][2][DC_SIZE];
    x264gpu_mb rec[16][2][3];                  // per wave and half: Q, left P, top P
    int progress[WF_MAX_ROWS];
    uint8_t alpha[52], beta[52], tc0[52][4], cqp[52];
};

// act: this half has a macroblock this step (per lane, uniform inside a half)
__device__ void deblock_mb_pair(const EncK &k, Deblock2Lds &L, int wave, int lane, int s, int mbx, int mby, bool act)
{
    const int hf = lane >> 5, l32 = lane & 31;
    uint8_t *lt = L.lt[wave][hf] + DL_ORG, *ct = L.ct[wave][hf] + DC_ORG;
    const x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    const int cmbx = act ? mbx : 0, cmby = act ? mby : 0;                     // inactive halves touch macroblock 0 harmlessly (loads only)
    const x264gpu_mb *gQ = mbs + cmby * k.mbw + cmbx;
    uint8_t *Y = rec_plane00(k, s) + (size_t)(cmby * 16) * k.rs + cmbx * 16;
    uint8_t *UV = rec_chroma00(k, s) + (size_t)(cmby * 8) * k.rs + cmbx * 16;

    // ---- one round trip: pixel neighbourhood + the three macroblock records into LDS ----
    for (int i = l32; i < 20 * 5; i += 32) {
        const int r = i / 5 - 4, c = (i % 5) * 4 - 4;
        *(uint32_t *)(lt + r * DL_STRIDE + c) = *(const uint32_t *)(Y + (long)r * k.rs + c);
    }
    for (int i = l32; i < 50; i += 32) {
        const int r = i / 5 - 2, c = (i % 5) * 4 - 4;
        *(uint32_t *)(ct + r * DC_STRIDE + c) = *(const uint32_t *)(UV + (long)r * k.rs + c);
    }
    for (int i = l32; i < 48; i += 32) {
        const int which = i >> 4, w = i & 15;
        const x264gpu_mb *src = which == 0 ? gQ : which == 1 ? (cmbx > 0 ? gQ - 1 : gQ) : (cmby > 0 ? gQ - k.mbw : gQ);
        ((uint32_t *)&L.rec[wave][hf][which])[w] = ((const uint32_t *)src)[w];
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    const x264gpu_mb *Q = &L.rec[wave][hf][0];
    // halves whose 32 edge segments all have bS 0 have nothing to filter or write back
    bool work = false;
    {
        const int dir = l32 >> 4, edge = (l32 >> 2) & 3, seg = l32 & 3;
        // under slice threads the filter stops at slice boundaries (disable_deblocking_filter_idc 2, as x264 codes them); plain --slices N: it does not
        const bool skip_edge = ((edge & 1) && Q->transform8x8) || (edge == 0 && (dir == 0 ? cmbx == 0 : (cmby == 0 || filter_stops_at_row(k, cmby))));
        bool any = false;
        if (act && !skip_edge) {
            const x264gpu_mb *P = edge == 0 ? &L.rec[wave][hf][dir == 0 ? 1 : 2] : Q;
            const int qbx = dir == 0 ? edge : seg, qby = dir == 0 ? seg : edge;
            const int pbx = dir == 0 ? (edge + 3) & 3 : seg, pby = dir == 0 ? seg : (edge + 3) & 3;
            any = edge_bs(P, pbx, pby, Q, qbx, qby, edge == 0, k) != 0;
        }
        const unsigned long long b = __ballot(any);
        work = hf ? (b >> 32) != 0 : (b & 0xffffffffull) != 0;
    }
    if (!__any(work)) return;
    const int qpq = Q->qp, qpcq = L.cqp[min(max(qpq + k.chroma_qp_offset, 0), 51)];
    for (int dir = 0; dir < 2; dir++)
        for (int edge = 0; edge < 4; edge++) {
            const x264gpu_mb *P = Q;
            bool go = work && !((edge & 1) && Q->transform8x8);
            if (edge == 0) {
                if (dir == 0) { go = go && cmbx != 0; P = &L.rec[wave][hf][1]; }
                else { go = go && cmby != 0 && !filter_stops_at_row(k, cmby); P = &L.rec[wave][hf][2]; }
            }
            if (__any(go)) {
                const int qpp = P->qp;
                const int qpav = (qpp + qpq + 1) >> 1, qpcav = (L.cqp[min(max(qpp + k.chroma_qp_offset, 0), 51)] + qpcq + 1) >> 1;
                const int ia = min(max(qpav + k.alpha_off, 0), 51), ib = min(max(qpav + k.beta_off, 0), 51);
                const int ica = min(max(qpcav + k.alpha_off, 0), 51), icb = min(max(qpcav + k.beta_off, 0), 51);
                if (go && l32 < 16) {
                    const int seg = l32 >> 2;
                    const int qbx = dir == 0 ? edge : seg, qby = dir == 0 ? seg : edge;
                    const int pbx = dir == 0 ? (edge + 3) & 3 : seg, pby = dir == 0 ? seg : (edge + 3) & 3;
                    const int bs = edge_bs(P, pbx, pby, Q, qbx, qby, edge == 0, k);
                    if (bs) {
                        const int tc0 = bs < 4 ? L.tc0[ia][bs - 1] : 0;
                        uint8_t *pix = dir == 0 ? lt + l32 * DL_STRIDE + edge * 4 : lt + edge * 4 * DL_STRIDE + l32;
                        filter_luma_line(pix, dir == 0 ? 1 : DL_STRIDE, L.alpha[ia], L.beta[ib], tc0, bs);
                    }
                } else if (go && !(edge & 1)) {
                    const int t = l32 - 16;            // vertical edge: chroma row 0..7 (U and V); horizontal: byte column 0..15
                    if (dir == 0 && t < 8) {
                        const int seg = t >> 1;
                        const int bs = edge_bs(P, (edge + 3) & 3, seg, Q, edge, seg, edge == 0, k);
                        if (bs) {
                            const int tc0 = bs < 4 ? L.tc0[ica][bs - 1] : 0;
                            uint8_t *pix = ct + t * DC_STRIDE + edge * 4;
                            filter_chroma_line(pix, 2, L.alpha[ica], L.beta[icb], tc0, bs);
                            filter_chroma_line(pix + 1, 2, L.alpha[ica], L.beta[icb], tc0, bs);
                        }
                    } else if (dir == 1) {
                        const int seg = t >> 2;
                        const int bs = edge_bs(P, seg, (edge + 3) & 3, Q, seg, edge, edge == 0, k);
                        if (bs) {
                            const int tc0 = bs < 4 ? L.tc0[ica][bs - 1] : 0;
                            uint8_t *pix = ct + (edge * 2) * DC_STRIDE + t;
                            filter_chroma_line(pix, DC_STRIDE, L.alpha[ica], L.beta[icb], tc0, bs);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }

    // ---- write back exactly what each macroblock may have modified ----
    if (work) {
        for (int i = l32; i < 64; i += 32) {   // own luma 16x16: 64 dwords
            const int r = i >> 2, c = (i & 3) * 4;
            *(uint32_t *)(Y + (long)r * k.rs + c) = *(const uint32_t *)(lt + r * DL_STRIDE + c);
        }
        if (cmbx > 0 && l32 < 16) *(uint32_t *)(Y + (long)l32 * k.rs - 4) = *(const uint32_t *)(lt + l32 * DL_STRIDE - 4);
        if (cmby > 0 && l32 >= 16 && l32 < 28) {
            const int i = l32 - 16, r = -3 + i / 4, c = (i & 3) * 4;
            *(uint32_t *)(Y + (long)r * k.rs + c) = *(const uint32_t *)(lt + r * DL_STRIDE + c);
        }
        {   // own chroma 8 rows x 16 bytes: 32 dwords
            const int r = l32 >> 2, c = (l32 & 3) * 4;
            *(uint32_t *)(UV + (long)r * k.rs + c) = *(const uint32_t *)(ct + r * DC_STRIDE + c);
        }
        if (cmbx > 0 && l32 < 8) *(uint32_t *)(UV + (long)l32 * k.rs - 4) = *(const uint32_t *)(ct + l32 * DC_STRIDE - 4);
        if (cmby > 0 && l32 >= 8 && l32 < 16) {
            const int i = l32 - 8, r = -2 + (i >> 2), c = (i & 3) * 4;
            *(uint32_t *)(UV + (long)r * k.rs + c) = *(const uint32_t *)(ct + r * DC_STRIDE + c);
        }
    }
}

// One workgroup per stream; wave w owns the row pairs w, w + 16, ...; the pair at step x filters (x, r0) and (x - 2, r0 + 1)
template <bool MWG>
__global__ __launch_bounds__(1024) void k_deblock2(EncK k)
{
    __shared__ __attribute__((aligned(16))) Deblock2Lds L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, s = blockIdx.x, nw = blockDim.x >> 6;      // scalar wave index
    const int gwave = MWG ? (int)blockIdx.y * nw + wave : wave, gstride = MWG ? (int)gridDim.y * nw : nw;
    for (int i = threadIdx.x; i < WF_MAX_ROWS; i += blockDim.x) L.progress[i] = 0;
    if (threadIdx.x < 52) {
        const int i = threadIdx.x;
        L.alpha[i] = d_alpha_table[i]; L.beta[i] = d_beta_table[i]; L.cqp[i] = d_chroma_qp_table[i];
        L.tc0[i][0] = d_tc0_table[i][0]; L.tc0[i][1] = d_tc0_table[i][1]; L.tc0[i][2] = d_tc0_table[i][2]; L.tc0[i][3] = 0;
    }
    __syncthreads();
    int *gprog = k.wf_progress + ((size_t)s * 2 + 1) * WFG_ROWS;     // MWG only
    auto pload = [&](int r) { if (MWG) return wfp_load<true>(gprog + r); return ((volatile int *)L.progress)[r]; };
    auto pstore = [&](int r, int v) { if (MWG) wfp_store<true>(gprog + r, v); else ((volatile int *)L.progress)[r] = v; };
    const int hf = lane >> 5;
    for (int r0 = 2 * gwave; r0 < k.mbh; r0 += 2 * gstride) {
        const bool row1 = r0 + 1 < k.mbh;
        for (int x = 0; x < k.mbw + 2; x++) {
            if (x < k.mbw && r0 > 0) {                                             // the upper row of the pair depends on the previous pair
                const int need = min(x + 2, k.mbw);
                while (pload(r0 - 1) < need) __builtin_amdgcn_s_sleep(2);
                wfp_acquire<MWG>();
            }
            const int mx = hf ? x - 2 : x;
            const bool act = hf ? (row1 && x >= 2) : x < k.mbw;
            deblock_mb_pair(k, L, wave, lane, s, mx, r0 + hf, act);
            wfp_release<MWG>();
            // the pair's lower row feeds the next pair's upper row; with an odd row count the upper row is the last one
            if (lane == 0) { pstore(r0, min(x + 1, k.mbw)); if (row1 && x >= 2) pstore(r0 + 1, x - 1); }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");               // this wave's lower row reads what its upper row just wrote
        }
    }
}

// chroma border expansion of the padded NV12 reference (frame_expand_border for the chroma plane)
__global__ __launch_bounds__(256) void k_chroma_border(EncK k)
{
    const int s = blockIdx.z;
    const int cw = k.cw / 2, chh = k.ch / 2;
    const int x = blockIdx.x * 256 + threadIdx.x - CPAD, y = blockIdx.y - CPAD;   // chroma sample coords
    if (x >= cw + CPAD) return;
    if (x >= 0 && x < cw && y >= 0 && y < chh) return;
    uint8_t *uv = rec_chroma00(k, s);
    const int sx = min(max(x, 0), cw - 1), sy = min(max(y, 0), chh - 1);
    *(uint16_t *)(uv + (long)y * k.rs + 2 * x) = *(const uint16_t *)(uv + (long)sy * k.rs + 2 * sx);
}

}  // namespace x264gpu

// k_intra2.cuh — intra macroblock analysis + encode on reconstructed neighbours (A5 + A6-A8), band version.
//
// One workgroup per stream; a wavefront owns a BAND of four consecutive macroblock rows and keeps one macroblock of
// each row in flight ("slots", one DPP row of 16 lanes each).  Every row has its own cursor over the intra macroblocks
// of that row, so in P slices (sparse intra macroblocks) the slots stay full; inside a band a slot may take the
// macroblock at x only when the row above has passed x+1 (top-right neighbour), which for I slices yields the 2:1
// wavefront.  Bands hand over through LDS progress counters with workgroup-scope fences, as in k_intra.
//
// Per step:  phase A (per active slot, whole wave, Z layout): neighbours -> LDS tiles, Intra16x16 decision
//            phase B (the four slots side by side): Intra8x8 then Intra4x4 analysis + encode — the two serial
//                    block loops that dominate the cost; a slot's 16 lanes split the modes between them
//                    (8x8: two 8-lane halves x 5 passes, R8 layout; 4x4: four quads x 3 passes)
//            phase C (per active slot, whole wave): final choice, record, Intra16x16 encode, chroma
// Same results as k_intra / oracle intra_mb, bit-exact.
#pragma once
#ifndef X264GPU_I2_WAVES
#define X264GPU_I2_WAVES 8           // wavefronts per workgroup (each owns bands w, w + WAVES, ...)
#endif
#include "k_intra.cuh"

namespace x264gpu {

constexpr int I2_WAVES = X264GPU_I2_WAVES;               // wavefronts per workgroup for many-stream launches
constexpr int I2_WAVES_MWG = 4;                          // multi-workgroup mode: one wave per SIMD
constexpr int I2_WAVES_FEW = 16;                         // ... and when few streams are in flight (latency over footprint)
struct SlotLds {
    __attribute__((aligned(8))) uint8_t tile[IT_SIZE];
    __attribute__((aligned(8))) uint8_t tile8[IT_SIZE];
    __attribute__((aligned(8))) uint8_t src[256];       // source macroblock, stride 16
    __attribute__((aligned(8))) int16_t lv8[256];       // Intra8x8 levels (interleaved 4x4 form) until the final choice
    uint8_t nb[NB_SIZE];
    uint8_t U[U_SIZE];
    uint8_t U8[U8_SIZE];
    uint8_t line[32];                                   // raw 8x8 reference line l7..l0, tl, t0..t15
    uint8_t cnb[2][CNB_SIZE];
    uint8_t modes[16], modes8[16], nmodes[8];
    int info[12];                                       // see enum below
    // adaptive quantisation: this macroblock's quantiser-dependent values (filled per macroblock in phase A; unused otherwise)
    struct { Q4 ql, qc; Q8 q8; int qp, lambda; } q;
};
enum { SI_BEST16 = 0, SI_MODE16, SI_COST8, SI_DONE8, SI_NNZ8, SI_CBP8, SI_COST4, SI_NNZ4, SI_MBX, SI_MBY };
template <int NW>
struct Intra2LdsT {
    SlotLds slot[NW][4];
    __attribute__((aligned(8))) uint8_t pred8tab[9 * 64];
    __attribute__((aligned(4))) uint8_t pred4tab[9 * 16];
    int progress[160];
};

// ---- phase A: neighbours of macroblock (mbx,mby) into the slot's LDS, Intra16x16 decision -------------------------
template <bool AQ>
__device__ __forceinline__ void i2_phase_a(const EncK &k, SlotLds &S, int lane, int s, int mbx, int mby)
{
    uint8_t *tile = S.tile + IT_ORG, *tile8 = S.tile8 + IT_ORG, *nb = S.nb;
    const int mbi = mby * k.mbw + mbx, px = mbx * 16, py = mby * 16;
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    const uint8_t *fenc = k.fenc_y + (size_t)s * k.fency_bytes + (size_t)py * k.fs + px;
    uint8_t *rec = rec_plane00(k, s) + (size_t)py * k.rs + px;
    const bool left = mbx > 0, top = mby > 0, topright = top && mbx + 1 < k.mbw;
    const int zx = z_x0(lane), zy = z_y(lane);
    if (AQ) {
        // this macroblock's quantiser and everything derived from it, from the per-quantiser tables into the slot
        const int qp = k.mbqp[(size_t)s * k.nmb + mbi], qpc = d_chroma_qp_table[min(max(qp + k.chroma_qp_offset, 0), 51)];
        const int *a = (const int *)&k.q4tab[qp * 4 + 0], *b = (const int *)&k.q4tab[qpc * 4 + 2], *c = (const int *)&k.q8tab[qp * 2 + 0];
        int *d = (int *)&S.q;
        constexpr int N4 = sizeof(Q4) / 4, N8 = sizeof(Q8) / 4;
        if (lane < N4) d[lane] = a[lane];
        else if (lane < 2 * N4) d[lane] = b[lane - N4];
        else if (lane < 2 * N4 + N8) d[lane] = c[lane - 2 * N4];
        else if (lane == 2 * N4 + N8) d[lane] = qp;
        else if (lane == 2 * N4 + N8 + 1) d[lane] = k.lambda_tab[qp];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    const int lambda = AQ ? S.q.lambda : k.lambda;
    // ---- neighbours: row -1 (x = -1..19) and column -1 into the tile; nb[] for the 16x16 predictors ----
        if (lane < 25) {                           // x = -1..23: the top-right 8 samples serve Intra_8x8 block 1
        const int x = lane - 1;
        const bool ok = top && (x >= 0 || left) && (x < 16 || topright);
        const uint8_t v = ok ? rec[-(long)k.rs + x] : 128;
        tile[-IT_STRIDE + x] = v;
        tile8[-IT_STRIDE + x] = v;
        if (lane < 21) nb[NB_TOP + x] = v;     // x = -1 lands on NB_TL
    } else if (lane >= 32 && lane < 48) {
        const int y = lane - 32;
        const uint8_t v = left ? rec[(long)y * k.rs - 1] : 128;
        tile[y * IT_STRIDE - 1] = v;
        tile8[y * IT_STRIDE - 1] = v;
        nb[NB_LEFT + y] = v;
    }
    // edge modes of the left / top macroblocks (DC unless that macroblock is I4x4 / I8x8), fetched once
    if (lane >= 48 && lane < 56) {
        const int i = lane - 48;
        int m = 2;
        if (i < 4 && left) { const x264gpu_mb *n = mbs + mbi - 1; if (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) m = n->i4_mode[blkidx_of(3, i)]; }
        if (i >= 4 && top) { const x264gpu_mb *n = mbs + mbi - k.mbw; if (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) m = n->i4_mode[blkidx_of(i - 4, 3)]; }
        S.nmodes[i] = (uint8_t)m;
    }
    const uint32_t cz = *(const uint32_t *)(fenc + (size_t)zy * k.fs + zx);
    *(uint32_t *)(S.src + zy * 16 + zx) = cz;          // source macroblock for the slot-parallel stages
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    // ---- intra 16x16 mode decision ----
    const Pred16 pp = pred16_setup(nb, lane);
    int best16 = 1 << 28, mode16 = 0;
    {
        int modes[4], n;
        if (left && top) { modes[0] = PRED16_V; modes[1] = PRED16_H; modes[2] = PRED16_DC; modes[3] = PRED16_P; n = 4; }
        else if (left) { modes[0] = PRED16_H; modes[1] = PRED16_DC_LEFT; n = 2; }
        else if (top) { modes[0] = PRED16_V; modes[1] = PRED16_DC_TOP; n = 2; }
        else { modes[0] = PRED16_DC_128; n = 1; }
        for (int i = 0; i < n; i++) {
            const int m = modes[i], sig = m > PRED16_P ? PRED16_DC : m;
            const int c = wave_sum(satd4_half(cz, pred16_row4(nb, pp, m, zx, zy), lane)) + lambda * bs_size_ue(sig);
            if (c < best16) { best16 = c; mode16 = m; }
        }
    }

    if (lane == 0) { S.info[SI_BEST16] = best16; S.info[SI_MODE16] = mode16; S.info[SI_MBX] = mbx; S.info[SI_MBY] = mby; }
    if (lane < 16) S.modes8[lane] = 2;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// ---- phase C: final choice, record, Intra16x16 encode, chroma ---------------------------------------------------
template <bool AQ>
__device__ __forceinline__ void i2_phase_c(const EncK &k, SlotLds &S, int lane, int s, int mbx, int mby)
{
    uint8_t *tile = S.tile + IT_ORG, *tile8 = S.tile8 + IT_ORG, *nb = S.nb;
    const int mbi = mby * k.mbw + mbx, px = mbx * 16, py = mby * 16;
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    int16_t *lv = k.levels + ((size_t)s * k.nmb + mbi) * X264GPU_MB_LEVELS;
    uint8_t *rec = rec_plane00(k, s) + (size_t)py * k.rs + px;
    const bool left = mbx > 0, top = mby > 0;
    const int j = lane & 3, zx = z_x0(lane), zy = z_y(lane);
    const int qp = AQ ? S.q.qp : k.qp, lambda = AQ ? S.q.lambda : k.lambda;
    const Q4 &q_luma_intra = AQ ? S.q.ql : k.q_luma_intra, &q_chroma_intra = AQ ? S.q.qc : k.q_chroma_intra;
    const uint32_t cz = *(const uint32_t *)(S.src + zy * 16 + zx);
    const Pred16 pp = pred16_setup(nb, lane);
    const int best16 = S.info[SI_BEST16], mode16 = S.info[SI_MODE16];
    const int cost8 = S.info[SI_COST8], cost4 = S.info[SI_COST4], cbp8 = S.info[SI_CBP8];
    const unsigned nnz8 = (unsigned)S.info[SI_NNZ8], nnz4 = (unsigned)S.info[SI_NNZ4];
    const bool i8_done = S.info[SI_DONE8] != 0;
    // 16x16, then 4x4 if strictly cheaper, then 8x8 if strictly cheaper still (x264's COPY2_IF_LT chain)
    bool use_i4 = (k.partitions & 2) && cost4 < best16 && (!i8_done || cost4 <= cost8);
    const bool use_i8 = i8_done && cost8 < (use_i4 ? cost4 : best16);
    if (use_i8) use_i4 = false;
    x264gpu_mb recd;
    __builtin_memset(&recd, 0, sizeof(recd));
    recd.qp = (uint8_t)qp;
    for (int i = 0; i < 4; i++) recd.ref[i] = -1;
    if (k.slice_type == X264GPU_SLICE_P) { recd.aux[0] = mbs[mbi].aux[0]; recd.aux[1] = mbs[mbi].aux[1]; recd.aux[2] = mbs[mbi].aux[2]; }

    if (use_i8) {
        recd.type = X264GPU_MB_I8x8;
        recd.cost = cost8;
        recd.transform8x8 = 1;
        for (int b = 0; b < 16; b++) recd.i4_mode[b] = S.modes8[b];
        recd.nnz = nnz8;
        recd.cbp_luma = (uint8_t)cbp8;
        *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = *(const uint32_t *)(tile8 + zy * IT_STRIDE + zx);
        *(uint2 *)(lv + lane * 4) = *(const uint2 *)(S.lv8 + lane * 4);       // 256 luma levels, interleaved 4x4 form
        if (lane < 16) lv[X264GPU_LV_LUMA_DC + lane] = 0;
    } else if (use_i4) {
        recd.type = X264GPU_MB_I4x4;
        recd.cost = cost4;
        for (int b = 0; b < 16; b++) recd.i4_mode[b] = S.modes[b];
        recd.nnz = nnz4;
        for (int i8 = 0; i8 < 4; i8++) if ((nnz4 >> (4 * i8)) & 15) recd.cbp_luma |= 1 << i8;
        // reconstructed luma: tile -> frame
        *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = *(const uint32_t *)(tile + zy * IT_STRIDE + zx);
        if (lane < 16) lv[X264GPU_LV_LUMA_DC + lane] = 0;
    } else {
        // ---- x264_mb_encode_i16x16 ----
        recd.type = X264GPU_MB_I16x16;
        recd.cost = best16;
        recd.i16_mode = (uint8_t)(mode16 > PRED16_P ? PRED16_DC : mode16);
        int e[4], p[4], v[4];
        unpack4(cz, e); unpack4(pred16_row4(nb, pp, mode16, zx, zy), p);
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] = e[t] - p[t];
        dct4_quad(v, lane);
        const int dcv = v[0];                      // meaningful on j == 0 lanes
        if (j == 0) v[0] = 0;
        quant4_row(v, q_luma_intra, j);
        const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
        store_levels_scan(lv + (lane >> 2) * 16, v, j);
        dequant4_row(v, q_luma_intra, j);
        const unsigned long long bal = __ballot(nz && j == 0);
        unsigned acn = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) acn |= (unsigned)((bal >> (4 * b)) & 1) << b;
        // DC matrix in natural layout on every quad: lane row r = j, register c -> block (bx=c, by=r)
        int dc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) dc[c] = __shfl(dcv, 4 * blkidx_of(c, j));
        had4x4_quad(dc, lane);
#pragma unroll
        for (int c = 0; c < 4; c++) dc[c] = quant_one((dc[c] + 1) >> 1, q_luma_intra.mf[0] >> 1, q_luma_intra.bias[0] << 1);
        const bool nzdc = quad_or((dc[0] | dc[1] | dc[2] | dc[3]) != 0 ? 1 : 0) != 0;
        if (lane < 4) store_levels_scan(lv + X264GPU_LV_LUMA_DC, dc, j);
        had4x4_quad(dc, lane);
        {
            const int ls = q_luma_intra.dq[0], qb = qp / 6 - 6;
#pragma unroll
            for (int c = 0; c < 4; c++) dc[c] = dequant_one(dc[c], ls, qb);
        }
        // hand each block its DC: the value lives in register bx of quad-lane by
        {
            const int b = lane >> 2, bx = z_bx(b), by = z_by(b);
            int t0 = __shfl(dc[0], by), t1 = __shfl(dc[1], by), t2 = __shfl(dc[2], by), t3 = __shfl(dc[3], by);
            const int mine = bx == 0 ? t0 : bx == 1 ? t1 : bx == 2 ? t2 : t3;
            if (j == 0) v[0] = nzdc ? mine : 0;
        }
        idct4_quad(v, lane);
#pragma unroll
        for (int t = 0; t < 4; t++) v[t] += p[t];
        *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = pack4_clip(v);
        recd.nnz = acn | (nzdc ? 1u << 24 : 0);
        recd.cbp_luma = acn ? 15 : 0;
    }
    if (lane >= 16 && lane < 24) lv[408 + lane - 16] = 0;

    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from interleaving phases (it costs ~80 spilled VGPRs otherwise)
    // ---- chroma: mode decision + encode (lanes 0..31; plane = lane>>4) ----
    {
        const int c = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
        uint8_t *cnb = S.cnb[c];
        uint8_t *ruv = rec_chroma00(k, s) + (size_t)(mby * 8) * k.rs + px;
        // neighbour ring of both planes: lanes 0..8 top (x=-1..7) and 9..16 left, per plane on its own DPP row
        {
            const int t = lane & 15, pl = (lane >> 4) & 1;
            if (lane < 32) {
                if (t < 9) { const int x = t - 1; S.cnb[pl][CNB_TOP + x] = (top && (x >= 0 || left)) ? ruv[-(long)k.rs + 2 * x + pl] : 128; }
            } else {
                const int y = t & 7, pl2 = (t >> 3) & 1;
                if (lane < 48) S.cnb[pl2][CNB_LEFT + y] = left ? ruv[(long)y * k.rs - 2 + pl2] : 128;
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const PredC pc = predc_setup(cnb);
        const uint8_t *fuv = k.fenc_uv + (size_t)s * k.fencuv_bytes + (size_t)(mby * 8 + cyy) * k.fs + px + 2 * cx0;
        const uint2 fe = *(const uint2 *)fuv;
        const uint32_t cenc = nv12_pick(fe.x, fe.y, c);
        int modes[4], n;
        if (left && top) { modes[0] = PREDC_DC; modes[1] = PREDC_H; modes[2] = PREDC_V; modes[3] = PREDC_P; n = 4; }
        else if (left) { modes[0] = PREDC_DC_LEFT; modes[1] = PREDC_H; n = 2; }
        else if (top) { modes[0] = PREDC_DC_TOP; modes[1] = PREDC_V; n = 2; }
        else { modes[0] = PREDC_DC_128; n = 1; }
        int bestc = 1 << 28, bestm = 0;
        for (int i = 0; i < n; i++) {
            const int m = modes[i], sig = m > PREDC_P ? PREDC_DC : m;
            const int hs = satd4_half(cenc, predc_row4(cnb, pc, m, ci, j), lane);
            const int cst = wave_sum(lane < 32 ? hs : 0) + lambda * bs_size_ue(sig);
            if (cst < bestc) { bestc = cst; bestm = m; }
        }
        recd.chroma_mode = (uint8_t)(bestm > PREDC_P ? PREDC_DC : bestm);
        const uint32_t cpred = predc_row4(cnb, pc, bestm, ci, j);
        int cbp_chroma = 0;
        unsigned nn = recd.nnz;
        const uint32_t crec = chroma_residual(cenc, cpred, q_chroma_intra, false, false, lane, lv, nn, cbp_chroma);
        recd.nnz = nn;
        recd.cbp_chroma = (uint8_t)cbp_chroma;
        const uint32_t other = (uint32_t)__shfl_xor((int)crec, 16);
        if (lane < 16) {
            const uint32_t u = crec, w = other;
            uint2 o;
            o.x = (u & 0xff) | ((w & 0xff) << 8) | ((u & 0xff00) << 8) | ((w & 0xff00) << 16);
            o.y = ((u >> 16) & 0xff) | (((w >> 16) & 0xff) << 8) | ((u >> 24) << 16) | ((w >> 24) << 24);
            *(uint2 *)(ruv + (size_t)cyy * k.rs + 2 * cx0) = o;
        }
    }
    if (lane == 0) mbs[mbi] = recd;
}

// ---- phase B: Intra8x8 then Intra4x4 of the four slots side by side -----------------------------------------------
// lane = (slot, 16 lanes).  Values that are uniform inside a slot (costs, modes, availability) live in VGPRs here.
template <bool AQ, class LDS>
__device__ __forceinline__ void i2_phase_b(const EncK &k, LDS &L, SlotLds *slots, int lane, int s, unsigned active)
{
    const int sl = lane >> 4, l16 = lane & 15;
    SlotLds &S = slots[sl];
    const bool act = (active >> sl) & 1;
    const int mbx = S.info[SI_MBX], mby = S.info[SI_MBY], best16 = S.info[SI_BEST16];
    const int lambda = AQ ? S.q.lambda : k.lambda;
    const Q4 &q_luma_intra = AQ ? S.q.ql : k.q_luma_intra;
    const Q8 &q8_intra = AQ ? S.q.q8 : k.q8_intra;
    const bool left = mbx > 0, top = mby > 0, topright = top && mbx + 1 < k.mbw;

    // ================= Intra8x8: half h of the slot evaluates modes h, h+2, h+4, h+6 (+ mode 8 on half 0) =================
    int cost8 = 1 << 28, cbp8 = 0;
    unsigned nnz8 = 0;
    bool alive8 = false;
    if ((k.partitions & 4) && k.dct8x8) {
        const int h = (lane >> 3) & 1, r8 = lane & 7;
        uint8_t *tile8 = S.tile8 + IT_ORG, *U8 = S.U8;
        cost8 = lambda * 4;
        alive8 = true;
        for (int i8 = 0; i8 < 4; i8++) {
            const int x8 = i8 & 1, y8 = i8 >> 1;
            int avail = 0;
            if (x8 || left) avail |= AVAIL_LEFT;
            if (y8 || top) avail |= AVAIL_TOP;
            if ((x8 || left) && (y8 || top)) avail |= AVAIL_TOPLEFT;
            if (i8 == 0 ? top : i8 == 1 ? topright : i8 == 2) avail |= AVAIL_TOPRIGHT;
            const bool has_tl = avail & AVAIL_TOPLEFT;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int pm = i4_pred_mode(S.nmodes, mbx, mby, i8 * 4, S.modes8);
            uint8_t *bt = tile8 + y8 * 8 * IT_STRIDE + x8 * 8;
            // reference line l7..l0, tl, t0..t15 (two samples per lane), then the 1-2-1 filter and the F2/F3/DC entries
            const int k0 = l16, k1 = l16 + 16;
            int r0v, r1v = 0;
            {
                if (k0 < 8) r0v = bt[(7 - k0) * IT_STRIDE - 1];
                else if (k0 == 8) r0v = bt[-IT_STRIDE - 1];
                else r0v = bt[-IT_STRIDE + (k0 - 9)];
                if (k1 < 25) { int x = k1 - 9; if (!(avail & AVAIL_TOPRIGHT)) x = 7; r1v = bt[-IT_STRIDE + x]; }
                S.line[k0] = (uint8_t)r0v;
                if (k1 < 25) S.line[k1] = (uint8_t)r1v;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            int e0, e1 = 0;
            {
                const int a0 = (k0 == 0 || (k0 == 9 && !has_tl)) ? r0v : S.line[k0 - 1];
                const int b0 = (k0 == 7 && !has_tl) ? r0v : S.line[k0 + 1];
                e0 = (a0 + 2 * r0v + b0 + 2) >> 2;
                S.U8[U8_E + k0] = (uint8_t)e0;
                if (k1 < 25) {
                    const int a1 = S.line[k1 - 1], b1 = k1 == 24 ? r1v : S.line[k1 + 1];
                    e1 = (a1 + 2 * r1v + b1 + 2) >> 2;
                    S.U8[U8_E + k1] = (uint8_t)e1;
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            {
                const int ea0 = k0 == 0 ? e0 : U8[U8_E + k0 - 1], eb0 = U8[U8_E + k0 + 1];
                U8[U8_F3 + k0] = (uint8_t)((ea0 + 2 * e0 + eb0 + 2) >> 2);
                U8[U8_F2 + k0] = (uint8_t)((e0 + eb0 + 1) >> 1);
                if (k1 < 25) {
                    const int ea1 = U8[U8_E + k1 - 1], eb1 = k1 == 24 ? e1 : U8[U8_E + k1 + 1];
                    U8[U8_F3 + k1] = (uint8_t)((ea1 + 2 * e1 + eb1 + 2) >> 2);
                    if (k1 < 24) U8[U8_F2 + k1] = (uint8_t)((e1 + eb1 + 1) >> 1);
                }
                if (l16 == 15) {
                    int sl_ = 0, st = 0;
                    for (int i = 0; i < 8; i++) { sl_ += U8[U8_E + i]; st += U8[U8_E + 9 + i]; }
                    const bool l = avail & AVAIL_LEFT, t = avail & AVAIL_TOP;
                    U8[U8_DC] = (uint8_t)(l && t ? (st + sl_ + 8) >> 4 : l ? (sl_ + 4) >> 3 : t ? (st + 4) >> 3 : 128);
                }
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // source rows of this block in R8 layout
            const uint2 en = *(const uint2 *)(S.src + (y8 * 8 + r8) * 16 + x8 * 8);
            unsigned key = 0xffffffffu;
#pragma unroll 1
            for (int p = 0; p < 5; p++) {
                const int m = p < 4 ? h + 2 * p : 8;
                uint32_t plo, phi;
                pred8_row8(U8, L.pred8tab, m, r8, plo, phi);
                int hs = sa8d_r8_half(en.x, en.y, plo, phi, lane);
                hs += dpp<DPP_XOR1>(hs); hs += dpp<DPP_XOR2>(hs); hs += xor4(hs);
                const int c = ((2 * hs + 2) >> 2) + (m == pm ? 0 : 3 * lambda);
                const bool ok = pred4_mode_ok(m, avail) && (p < 4 || h == 0);
                key = min(key, ok ? (((unsigned)c << 4) | (unsigned)m) : 0xffffffffu);
            }
            { const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)key, 0x128, 0xf, 0xf, true); key = min(key, t); }   // other half (row_ror:8)
            const int bm = key & 15;
            cost8 += (int)(key >> 4);
            if (l16 < 4) S.modes8[i8 * 4 + l16] = (uint8_t)bm;
            if (i8 < 3 && cost8 > best16) alive8 = false;                 // cannot win any more (x264 breaks here)
            if (!__any(act && alive8)) break;
            // encode with the winning prediction (both halves do the same work; half 0 stores)
            uint32_t plo, phi;
            pred8_row8(U8, L.pred8tab, bm, r8, plo, phi);
            int e[8], p[8], v[8];
            unpack8(en.x, en.y, e); unpack8(plo, phi, p);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
            fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
            int mf[4], bs[4], dq[4];
            q8_row(q8_intra, r8, mf, bs, dq);
            unsigned mlo = 0, mhi = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                const int z = c_zigzag8_inv[r8 * 8 + i];
                if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                if (h == 0) S.lv8[(i8 * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)v[i];
            }
            mlo = group8_or(mlo); mhi = group8_or(mhi);
            const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
#pragma unroll
            for (int q = 0; q < 4; q++) nnz8 |= (mask & (0x1111111111111111ull << q)) ? 1u << (i8 * 4 + q) : 0u;
            if (mask) cbp8 |= 1 << i8;
            const int qb = q8_intra.qp / 6 - 6;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
            inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
            if (h == 0) {
                *(uint32_t *)(bt + r8 * IT_STRIDE) = pack4_clip8lo(v);
                *(uint32_t *)(bt + r8 * IT_STRIDE + 4) = pack4_clip8hi(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }

    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from interleaving phases (it costs ~80 spilled VGPRs otherwise)
    // ================= Intra4x4: quad q of the slot evaluates modes q, q+4 (+ mode 8 on quad 0) =================
    int cost4 = 1 << 28;
    unsigned nnz4 = 0;
    if (k.partitions & 2) {
        const int thr8 = alive8 ? cost8 : (1 << 28);
        const int q = (lane >> 2) & 3, j = lane & 3;
        uint8_t *tile = S.tile + IT_ORG;
        int16_t *lvp = k.levels + ((size_t)s * k.nmb + (act ? mby * k.mbw + mbx : 0)) * X264GPU_MB_LEVELS;
        const uint32_t *tab4 = (const uint32_t *)L.pred4tab;
        cost4 = lambda * (24 + 16);
        for (int b = 0; b < 16; b++) {
            const int bx = z_bx(b), by = z_by(b);
            const int avail = i4_avail(mbx, mby, k.mbw, b);
            const int pm = i4_pred_mode(S.nmodes, mbx, mby, b, S.modes);
            uint8_t *bt = tile + by * 4 * IT_STRIDE + bx * 4;
            pred4_build_u(S.U, bt, IT_STRIDE, avail, l16);
            const uint32_t en = *(const uint32_t *)(S.src + (by * 4 + j) * 16 + bx * 4);
            unsigned key = 0xffffffffu;
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const int m = q + 4 * p, mm = m < 9 ? m : 2;
                const uint32_t pr = pred4_row4(S.U, tab4[mm * 4 + j]);
                const int sat = quad_sum(satd4_half(en, pr, lane));
                const bool ok = m < 9 && pred4_mode_ok(mm, avail);
                key = min(key, ok ? (((unsigned)(sat + (m == pm ? 0 : 3 * lambda)) << 4) | (unsigned)m) : 0xffffffffu);
            }
            key = row16_min_u32(key);
            const int bm = key & 15;
            cost4 += (int)(key >> 4);
            if (l16 == 0) S.modes[b] = (uint8_t)bm;
            const uint32_t bp = pred4_row4(S.U, tab4[bm * 4 + j]);
            int e[4], p[4], v[4];
            unpack4(en, e); unpack4(bp, p);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = e[t] - p[t];
            dct4_quad(v, lane);
            quant4_row(v, q_luma_intra, j);
            const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
            if (act && q == 0) store_levels_scan(lvp + b * 16, v, j);
            dequant4_row(v, q_luma_intra, j);
            idct4_quad(v, lane);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] += p[t];
            if (q == 0) *(uint32_t *)(bt + j * IT_STRIDE) = pack4_clip(v);
            if (nz) nnz4 |= 1u << b;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (!__any(act && cost4 < best16 && cost4 <= thr8)) break;   // no slot's Intra4x4 can win any more (costs only grow)
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from interleaving phases (it costs ~80 spilled VGPRs otherwise)
    if (l16 == 0) {
        S.info[SI_COST8] = cost8; S.info[SI_DONE8] = alive8 ? 1 : 0; S.info[SI_NNZ8] = (int)nnz8; S.info[SI_CBP8] = cbp8;
        S.info[SI_COST4] = cost4; S.info[SI_NNZ4] = (int)nnz4;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// next intra macroblock of `row` at or after `from` (mbw if none); inter macroblocks were reconstructed by k_encode_inter
__device__ __forceinline__ int i2_next_intra(const EncK &k, const x264gpu_mb *mbs, int row, int from, int lane)
{
    if (k.slice_type == X264GPU_SLICE_I) return min(from, k.mbw);
    while (from < k.mbw) {
        const int base = from & ~63, x = base + lane;
        bool is_intra = false;
        if (x < k.mbw && x >= from) { const int t = mbs[row * k.mbw + x].type; is_intra = t != X264GPU_MB_P_L0 && t != X264GPU_MB_P_8x8; }
        const unsigned long long m = __ballot(is_intra);
        if (m) return base + __builtin_ctzll(m);
        from = base + 64;
    }
    return k.mbw;
}

#ifndef X264GPU_I2_OCC
#define X264GPU_I2_OCC 4          // waves per SIMD the register allocator targets (2 -> up to 256 VGPRs, 4 -> 128)
#endif
// ROWS = macroblock rows per band (slots in use): 4 fills the wave when many pictures are in flight; 1 gives every row of a lone
// picture its own wavefront (single-stream latency: four times as many bands walk the picture at once)
template <int NW, bool MWG, bool AQ, int ROWS = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(X264GPU_I2_OCC, 8))) void k_intra2(EncK kk)
{
    __shared__ __attribute__((aligned(16))) Intra2LdsT<NW> L;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, s = blockIdx.x;      // scalar wave index: see k_analyse_p
    // MWG: gridDim.y workgroups share the bands of stream s (band = global wave index, then + all waves)
    const int gwave = MWG ? (int)blockIdx.y * NW + wave : wave, gstride = MWG ? (int)gridDim.y * NW : NW;
    for (int i = threadIdx.x; i < 160; i += NW * 64) L.progress[i] = 0;
    if (threadIdx.x < 144) ((uint32_t *)L.pred8tab)[threadIdx.x] = ((const uint32_t *)c_pred8_table)[threadIdx.x];
    if (threadIdx.x < 36) ((uint32_t *)L.pred4tab)[threadIdx.x] = ((const uint32_t *)c_pred4_table.t)[threadIdx.x];
    if (lane < 4 * 12) L.slot[wave][lane / 12].info[lane % 12] = 0;
    __syncthreads();
    const EncK &k = kk;
    const x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    int *gprog = kk.wf_progress + (size_t)s * 2 * WFG_ROWS;          // MWG only; the LDS counters are addressed as LDS (no generic pointer)
    auto pload = [&](int r) { if (MWG) return wfp_load<true>(gprog + r); return ((volatile int *)L.progress)[r]; };
    auto pstore = [&](int r, int v) { if (MWG) wfp_store<true>(gprog + r, v); else ((volatile int *)L.progress)[r] = v; };
    SlotLds *slots = L.slot[wave];
    unsigned long long tA = 0, tB = 0, tC = 0, tW = 0, nstep = 0, nslot = 0, t_begin = k.dbg ? clock64() : 0;
    const int mbw = k.mbw, nbands = (k.mbh + ROWS - 1) / ROWS;
    for (int band = gwave; band < nbands; band += gstride) {
        const int r0 = band * ROWS;
        int x[4], cur[4];                      // next intra macroblock / macroblocks completed, per row of the band
#pragma unroll
        for (int i = 0; i < 4; i++) { x[i] = i < ROWS && r0 + i < k.mbh ? i2_next_intra(k, mbs, r0 + i, 0, lane) : mbw; cur[i] = x[i]; }
        wfp_release<MWG>();
        if (lane < ROWS && r0 + lane < k.mbh) pstore(r0 + lane, lane == 0 ? cur[0] : lane == 1 ? cur[1] : lane == 2 ? cur[2] : cur[3]);
        for (;;) {
            unsigned act = 0;
            bool all_done = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (x[i] >= mbw) continue;
                all_done = false;
                const int need = min(x[i] + 2, mbw);           // the row above must have passed the top-right neighbour
                const int above = i == 0 ? (r0 == 0 ? mbw : pload(r0 - 1)) : cur[i - 1];
                if (above >= need) act |= 1u << i;
            }
            if (all_done) break;
            if (!act) { const unsigned long long w0 = k.dbg ? clock64() : 0; __builtin_amdgcn_s_sleep(2); if (k.dbg) tW += clock64() - w0; continue; }
            const unsigned long long c0 = k.dbg ? clock64() : 0;
            wfp_acquire<MWG>();
#pragma unroll 1
            for (int i = 0; i < ROWS; i++)            // one copy of the phase code (instruction-cache footprint), slot chosen at run time
                if (act >> i & 1) { i2_phase_a<AQ>(k, slots[i], lane, s, i == 0 ? x[0] : i == 1 ? x[1] : i == 2 ? x[2] : x[3], r0 + i); __builtin_amdgcn_sched_barrier(0); }
            const unsigned long long c1 = k.dbg ? clock64() : 0;
            __builtin_amdgcn_sched_barrier(0);
            i2_phase_b<AQ>(k, L, slots, lane, s, act);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long c2 = k.dbg ? clock64() : 0;
#pragma unroll 1
            for (int i = 0; i < ROWS; i++)
                if (act >> i & 1) { __builtin_amdgcn_sched_barrier(0); i2_phase_c<AQ>(k, slots[i], lane, s, i == 0 ? x[0] : i == 1 ? x[1] : i == 2 ? x[2] : x[3], r0 + i); }
            if (k.dbg) { const unsigned long long c3 = clock64(); tA += c1 - c0; tB += c2 - c1; tC += c3 - c2; nstep++; nslot += __builtin_popcount(act); }
            wfp_release<MWG>();
#pragma unroll
            for (int i = 0; i < 4; i++) if (act >> i & 1) { x[i] = i2_next_intra(k, mbs, r0 + i, x[i] + 1, lane); cur[i] = x[i]; }
            if (lane < ROWS && r0 + lane < k.mbh) pstore(r0 + lane, lane == 0 ? cur[0] : lane == 1 ? cur[1] : lane == 2 ? cur[2] : cur[3]);
        }
    }
    if (k.dbg && lane == 0) {
        unsigned long long *d = k.dbg + ((size_t)s * 16 + wave) * 16;   // [stream][16 wave slots][16]
        d[0] = tW; d[1] = tA + tB + tC; d[2] = nslot; d[3] = clock64() - t_begin;
        d[8] = tA; d[9] = tB; d[10] = tC; d[11] = nstep;
    }
}

}  // namespace x264gpu

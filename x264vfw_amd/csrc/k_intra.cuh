// k_intra.cuh — intra macroblock analysis + encode on reconstructed neighbours (A5 + A6-A8), run as a
// 2-D wavefront: ONE 1024-thread workgroup per stream walks the anti-diagonals d = x + 2y; its 16
// wavefronts take the macroblocks of a diagonal, a workgroup barrier separates diagonals (no
// inter-workgroup hand-off, so no agent-scope fences are needed).  Intra 4x4 evaluates the nine modes
// of a block in parallel: one quad of lanes per mode (table-driven predictors, intra.cuh).
// Restates oracle/encoder.c intra_mb bit-exactly.
#pragma once
#include "enc_common.cuh"
#include "intra8.cuh"

namespace x264gpu {

constexpr int IT_STRIDE = 32;                 // luma tile stride (bytes)
constexpr int IT_ORG = IT_STRIDE + 4;         // offset of sample (0,0): row -1 and columns -4..-1 precede it
constexpr int IT_SIZE = 17 * IT_STRIDE + 8;

struct IntraLds {
    uint8_t tile[16][IT_SIZE];
    uint8_t nb[16][NB_SIZE];
    uint8_t U[16][U_SIZE];
    uint8_t cnb[16][2][CNB_SIZE];
    uint8_t modes[16][16];
    // Intra_8x8 works in its own tile / level buffer so the 4x4 candidate's results survive until the final choice
    __attribute__((aligned(8))) uint8_t tile8[16][IT_SIZE];
    __attribute__((aligned(8))) int16_t lv8[16][256];
    uint8_t U8[16][U8_SIZE];
    uint8_t modes8[16][16];
    __attribute__((aligned(8))) uint8_t pred8tab[9 * 64];
    uint8_t nmodes[16][8];      // neighbour macroblocks' edge modes: [0..3] left column (by 0..3), [4..7] top row (bx 0..3)
    int progress[160];
    unsigned long long sect[16][8];   // diagnostics: per-wave section cycle accumulators
};

__device__ __forceinline__ int blkidx_of(int bx, int by) { return ((by >> 1) * 2 + (bx >> 1)) * 4 + (by & 1) * 2 + (bx & 1); }

// neighbour availability of 4x4 block b (oracle i4_avail)
__device__ __forceinline__ int i4_avail(int mbx, int mby, int mbw, int b)
{
    const int bx = z_bx(b), by = z_by(b);
    int a = 0;
    if (bx > 0 || mbx > 0) a |= AVAIL_LEFT;
    if (by > 0 || mby > 0) a |= AVAIL_TOP;
    if ((bx > 0 || mbx > 0) && (by > 0 || mby > 0)) a |= AVAIL_TOPLEFT;
    if (by == 0) { if (mby > 0 && (bx < 3 || mbx + 1 < mbw)) a |= AVAIL_TOPRIGHT; }
    else if (bx < 3 && blkidx_of(bx + 1, by - 1) < b) a |= AVAIL_TOPRIGHT;
    return a;
}

// predicted intra 4x4 mode (8.3.1.1; oracle i4_pred_mode)
__device__ __forceinline__ int i4_pred_mode(const uint8_t *nm, int mbx, int mby, int b, const uint8_t *cur)
{
    const int bx = z_bx(b), by = z_by(b);
    int ma, mb_;
    if (bx > 0) ma = cur[blkidx_of(bx - 1, by)];
    else if (mbx > 0) ma = nm[by];
    else return 2;
    if (by > 0) mb_ = cur[blkidx_of(bx, by - 1)];
    else if (mby > 0) mb_ = nm[4 + bx];
    else return 2;
    return min(ma, mb_);
}

__device__ void intra_mb_wave(const EncK &k, IntraLds &L, int wave, int lane, int s, int mbx, int mby, uint32_t t4)
{
    uint8_t *tile = L.tile[wave] + IT_ORG;     // sample (0,0)
    uint8_t *nb = L.nb[wave];
    uint8_t *U = L.U[wave];
    uint8_t *m4 = L.modes[wave];
    const int mbi = mby * k.mbw + mbx, px = mbx * 16, py = mby * 16;
    x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    int16_t *lv = k.levels + ((size_t)s * k.nmb + mbi) * X264GPU_MB_LEVELS;
    const uint8_t *fenc = k.fenc_y + (size_t)s * k.fency_bytes + (size_t)py * k.fs + px;
    uint8_t *rec = rec_plane00(k, s) + (size_t)py * k.rs + px;
    const bool left = mbx > 0, top = mby > 0, topright = top && mbx + 1 < k.mbw;
    const int j = lane & 3, zx = z_x0(lane), zy = z_y(lane);
    const int qp = k.qp, lambda = k.lambda;
    unsigned long long ts0 = k.dbg ? clock64() : 0;
#define SECT(i) do { if (k.dbg) { unsigned long long t_ = clock64(); if (lane == 0) L.sect[wave][i] += t_ - ts0; ts0 = t_; } } while (0)

    // ---- neighbours: row -1 (x = -1..19) and column -1 into the tile; nb[] for the 16x16 predictors ----
    uint8_t *tile8 = L.tile8[wave] + IT_ORG;
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
        L.nmodes[wave][i] = (uint8_t)m;
    }
    const uint32_t cz = *(const uint32_t *)(fenc + (size_t)zy * k.fs + zx);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);

    SECT(0);
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

    SECT(1);
    // ---- intra 8x8 (oracle intra_mb, I8x8 branch): R8 layout, lane = (mode group, row); eight modes per pass, the
    //      ninth (HU) in a second pass on group 0.  Candidate results stay in LDS (tile8 / lv8) until the final choice ----
    // The final choice (16x16, then 4x4 if strictly cheaper, then 8x8 if strictly cheaper still: x264's COPY2_IF_LT chain)
    // does not depend on the order of the two analyses; running 8x8 first lets the 16-block 4x4 loop stop as soon as it
    // can no longer win.
    bool use_i8 = false, i8_done = false;
    unsigned nnz8 = 0;
    int cost8 = 0, cbp8 = 0;
    if ((k.partitions & 4) && k.dct8x8) {
        const int cur = best16;
        uint8_t *m8 = L.modes8[wave], *U8 = L.U8[wave];
        int16_t *lv8 = L.lv8[wave];
        const int g = lane >> 3, r8 = lane & 7;
        if (lane < 16) m8[lane] = 2;
        cost8 = lambda * 4;
        bool done = true;
        for (int i8 = 0; i8 < 4; i8++) {
            const int x8 = i8 & 1, y8 = i8 >> 1;
            int avail = 0;
            if (x8 || left) avail |= AVAIL_LEFT;
            if (y8 || top) avail |= AVAIL_TOP;
            if ((x8 || left) && (y8 || top)) avail |= AVAIL_TOPLEFT;
            if (i8 == 0 ? top : i8 == 1 ? topright : i8 == 2) avail |= AVAIL_TOPRIGHT;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const int pm = i4_pred_mode(L.nmodes[wave], mbx, mby, i8 * 4, m8);
            uint8_t *bt = tile8 + y8 * 8 * IT_STRIDE + x8 * 8;
            pred8_build_u(U8, bt, IT_STRIDE, avail, lane);
            // source rows of this 8x8 block in R8 layout (every mode group gets a copy)
            const int src = i8 * 16 + (r8 >> 2) * 8 + (r8 & 3);
            const uint32_t elo = (uint32_t)__shfl((int)cz, src), ehi = (uint32_t)__shfl((int)cz, src + 4);
            uint32_t p1lo, p1hi, p2lo, p2hi;
            unsigned key;
            {   // modes 0..7, one per group
                pred8_row8(U8, L.pred8tab, g, r8, p1lo, p1hi);
                int h = sa8d_r8_half(elo, ehi, p1lo, p1hi, lane);
                h += dpp<DPP_XOR1>(h); h += dpp<DPP_XOR2>(h); h += xor4(h);
                const int c = ((2 * h + 2) >> 2) + (g == pm ? 0 : 3 * lambda);
                key = pred4_mode_ok(g, avail) ? (((unsigned)c << 4) | (unsigned)g) : 0xffffffffu;
            }
            {   // mode 8 on group 0
                pred8_row8(U8, L.pred8tab, 8, r8, p2lo, p2hi);
                int h = sa8d_r8_half(elo, ehi, p2lo, p2hi, lane);
                h += dpp<DPP_XOR1>(h); h += dpp<DPP_XOR2>(h); h += xor4(h);
                const int c = ((2 * h + 2) >> 2) + (8 == pm ? 0 : 3 * lambda);
                const unsigned k2 = (g == 0 && pred4_mode_ok(8, avail)) ? (((unsigned)c << 4) | 8u) : 0xffffffffu;
                key = min(key, k2);
            }
            key = wave_min_u32(key);
            const int bm = key & 15;
            cost8 += (int)(key >> 4);
            if (lane < 4) m8[i8 * 4 + lane] = (uint8_t)bm;
            if (i8 < 3 && cost8 > cur) { done = false; break; }      // cannot win any more
            // winning prediction to every group, then encode (all groups do the same work; group 0 stores)
            const uint32_t plo = bm == 8 ? (uint32_t)__shfl((int)p2lo, r8) : (uint32_t)__shfl((int)p1lo, bm * 8 + r8);
            const uint32_t phi = bm == 8 ? (uint32_t)__shfl((int)p2hi, r8) : (uint32_t)__shfl((int)p1hi, bm * 8 + r8);
            int e[8], p[8], v[8];
            unpack8(elo, ehi, e); unpack8(plo, phi, p);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = e[i] - p[i];
            fwd8_1d(v); transpose8(v, lane); fwd8_1d(v); transpose8(v, lane);
            int mf[4], bs[4], dq[4];
            q8_row(k.q8_intra, r8, mf, bs, dq);
            unsigned mlo = 0, mhi = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v[i] = quant_one(v[i], mf[i & 3], bs[i & 3]);
                const int z = c_zigzag8_inv[r8 * 8 + i];
                if (v[i]) { if (z < 32) mlo |= 1u << z; else mhi |= 1u << (z - 32); }
                if (g == 0) lv8[(i8 * 4 + (z & 3)) * 16 + (z >> 2)] = (int16_t)v[i];
            }
            mlo = group8_or(mlo); mhi = group8_or(mhi);
            const unsigned long long mask = ((unsigned long long)mhi << 32) | mlo;
#pragma unroll
            for (int q = 0; q < 4; q++) nnz8 |= (mask & (0x1111111111111111ull << q)) ? 1u << (i8 * 4 + q) : 0u;   // group 0's value is read below
            if (mask) cbp8 |= 1 << i8;
            const int qb = k.q8_intra.qp / 6 - 6;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = dequant_one(v[i], dq[i & 3], qb);
            inv8_1d(v); transpose8(v, lane); inv8_1d(v); transpose8(v, lane);
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = ((v[i] + 32) >> 6) + p[i];
            if (g == 0) {
                *(uint32_t *)(bt + r8 * IT_STRIDE) = pack4_clip8lo(v);
                *(uint32_t *)(bt + r8 * IT_STRIDE + 4) = pack4_clip8hi(v);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);
        nnz8 = (unsigned)__builtin_amdgcn_readfirstlane((int)nnz8);
        cbp8 = __builtin_amdgcn_readfirstlane(cbp8);
        i8_done = done;
    }

    SECT(2);
    // ---- intra 4x4: nine modes per block in parallel, blocks in coding order ----
    bool use_i4 = false;
    const int thr8 = i8_done ? cost8 : (1 << 28);
    unsigned nnz = 0;
    int cost4 = 0;
    if (k.partitions & 2) {
        cost4 = lambda * (24 + 16);
        for (int b = 0; b < 16; b++) {
            const int bx = z_bx(b), by = z_by(b);
            const int avail = i4_avail(mbx, mby, k.mbw, b);
            const int pm = i4_pred_mode(L.nmodes[wave], mbx, mby, b, m4);
            uint8_t *bt = tile + by * 4 * IT_STRIDE + bx * 4;
            pred4_build_u(U, bt, IT_STRIDE, avail, lane);
            const int m = lane >> 2;
            const bool ok = m < 9 && pred4_mode_ok(m, avail);
            const uint32_t pr = pred4_row4(U, t4);
            const uint32_t en = (uint32_t)__shfl((int)cz, b * 4 + j);
            int e[4], p[4];
            unpack4(en, e);
            const int sat = quad_sum(satd4_half(en, pr, lane));
            unsigned key = ok ? (((unsigned)(sat + (m == pm ? 0 : 3 * lambda)) << 4) | (unsigned)m) : 0xffffffffu;
            key = wave_min_u32(key);
            const int bm = key & 15;
            cost4 += (int)(key >> 4);
            if (lane == 0) m4[b] = (uint8_t)bm;
            // encode the block with the winning prediction (every quad does the same work)
            const uint32_t bp = (uint32_t)__shfl((int)pr, bm * 4 + j);
            int v[4];
            unpack4(bp, p);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] = e[t] - p[t];
            dct4_quad(v, lane);
            quant4_row(v, k.q_luma_intra, j);
            const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
            if (lane < 4) store_levels_scan(lv + b * 16, v, j);
            dequant4_row(v, k.q_luma_intra, j);
            idct4_quad(v, lane);
#pragma unroll
            for (int t = 0; t < 4; t++) v[t] += p[t];
            if (lane < 4) *(uint32_t *)(bt + j * IT_STRIDE) = pack4_clip(v);
            if (nz) nnz |= 1u << b;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (cost4 >= best16 || cost4 > thr8) break;            // intra 4x4 can no longer win (costs only grow)
        }
        use_i4 = cost4 < best16 && cost4 <= thr8;
    }
    use_i8 = i8_done && cost8 < (use_i4 ? cost4 : best16);
    if (use_i8) use_i4 = false;

    SECT(3);
    x264gpu_mb recd;
    __builtin_memset(&recd, 0, sizeof(recd));
    recd.qp = (uint8_t)qp;
    for (int i = 0; i < 4; i++) recd.ref[i] = -1;
    if (k.slice_type == X264GPU_SLICE_P) { recd.aux[0] = mbs[mbi].aux[0]; recd.aux[1] = mbs[mbi].aux[1]; recd.aux[2] = mbs[mbi].aux[2]; }

    if (use_i8) {
        recd.type = X264GPU_MB_I8x8;
        recd.cost = cost8;
        recd.transform8x8 = 1;
        for (int b = 0; b < 16; b++) recd.i4_mode[b] = L.modes8[wave][b];
        recd.nnz = nnz8;
        recd.cbp_luma = (uint8_t)cbp8;
        *(uint32_t *)(rec + (size_t)zy * k.rs + zx) = *(const uint32_t *)(tile8 + zy * IT_STRIDE + zx);
        *(uint2 *)(lv + lane * 4) = *(const uint2 *)(L.lv8[wave] + lane * 4);       // 256 luma levels, interleaved 4x4 form
        if (lane < 16) lv[X264GPU_LV_LUMA_DC + lane] = 0;
    } else if (use_i4) {
        recd.type = X264GPU_MB_I4x4;
        recd.cost = cost4;
        for (int b = 0; b < 16; b++) recd.i4_mode[b] = m4[b];
        recd.nnz = nnz;
        for (int i8 = 0; i8 < 4; i8++) if ((nnz >> (4 * i8)) & 15) recd.cbp_luma |= 1 << i8;
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
        quant4_row(v, k.q_luma_intra, j);
        const bool nz = quad_or((v[0] | v[1] | v[2] | v[3]) != 0 ? 1 : 0) != 0;
        store_levels_scan(lv + (lane >> 2) * 16, v, j);
        dequant4_row(v, k.q_luma_intra, j);
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
        for (int c = 0; c < 4; c++) dc[c] = quant_one((dc[c] + 1) >> 1, k.q_luma_intra.mf[0] >> 1, k.q_luma_intra.bias[0] << 1);
        const bool nzdc = quad_or((dc[0] | dc[1] | dc[2] | dc[3]) != 0 ? 1 : 0) != 0;
        if (lane < 4) store_levels_scan(lv + X264GPU_LV_LUMA_DC, dc, j);
        had4x4_quad(dc, lane);
        {
            const int ls = k.q_luma_intra.dq[0], qb = qp / 6 - 6;
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
    SECT(4);

    // ---- chroma: mode decision + encode (lanes 0..31; plane = lane>>4) ----
    {
        const int c = (lane >> 4) & 1, ci = (lane >> 2) & 3, cx0 = (ci & 1) * 4, cyy = (ci >> 1) * 4 + j;
        uint8_t *cnb = L.cnb[wave][c];
        uint8_t *ruv = rec_chroma00(k, s) + (size_t)(mby * 8) * k.rs + px;
        // neighbour ring of both planes: lanes 0..8 top (x=-1..7) and 9..16 left, per plane on its own DPP row
        {
            const int t = lane & 15, pl = (lane >> 4) & 1;
            if (lane < 32) {
                if (t < 9) { const int x = t - 1; L.cnb[wave][pl][CNB_TOP + x] = (top && (x >= 0 || left)) ? ruv[-(long)k.rs + 2 * x + pl] : 128; }
            } else {
                const int y = t & 7, pl2 = (t >> 3) & 1;
                if (lane < 48) L.cnb[wave][pl2][CNB_LEFT + y] = left ? ruv[(long)y * k.rs - 2 + pl2] : 128;
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
        const uint32_t crec = chroma_residual(cenc, cpred, k.q_chroma_intra, false, false, lane, lv, nn, cbp_chroma);
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
    SECT(5);
    if (lane == 0) mbs[mbi] = recd;
#undef SECT
}

__global__ __launch_bounds__(1024) void k_intra(EncK k)
{
    __shared__ __attribute__((aligned(16))) IntraLds L;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, s = blockIdx.x;
    // this lane's four table entries for (mode = lane>>2, row = lane&3); lanes of modes >= 9 are idle
    const uint32_t t4 = (lane >> 2) < 9 ? ((const uint32_t *)c_pred4_table.t)[lane] : 0x01010101u * U_DC;
    for (int i = threadIdx.x; i < 160; i += 1024) L.progress[i] = 0;
    if (threadIdx.x < 128) ((unsigned long long *)L.sect)[threadIdx.x] = 0;
    if (threadIdx.x < 144) ((uint32_t *)L.pred8tab)[threadIdx.x] = ((const uint32_t *)c_pred8_table)[threadIdx.x];
    __syncthreads();
    const x264gpu_mb *mbs = k.mb + (size_t)s * k.nmb;
    volatile int *progress = L.progress;
    unsigned long long t_wait = 0, t_work = 0, n_mb = 0, t_begin = k.dbg ? clock64() : 0;
    // wave w owns macroblock rows w, w+16, ...; an intra macroblock at x needs row-1 complete up to x+1
    // (top-right neighbour).  Inter macroblocks were reconstructed by k_encode_inter and are skipped.
    for (int row = wave; row < k.mbh; row += 16) {
        for (int x0 = 0; x0 < k.mbw; x0 += 64) {
            bool is_intra = false;
            if (x0 + lane < k.mbw) { const int t = mbs[row * k.mbw + x0 + lane].type; is_intra = k.slice_type == X264GPU_SLICE_I || (t != X264GPU_MB_P_L0 && t != X264GPU_MB_P_8x8); }
            unsigned long long todo = __ballot(is_intra);
            const int chunk_end = min(x0 + 64, k.mbw);
            // everything left of the next intra macroblock of this row is already reconstructed
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) progress[row] = todo ? x0 + __builtin_ctzll(todo) : chunk_end;
            while (todo) {
                const int x = x0 + __builtin_ctzll(todo);
                todo &= todo - 1;
                const unsigned long long t0 = k.dbg ? clock64() : 0;
                if (row > 0) {
                    const int need = min(x + 2, k.mbw);
                    while (progress[row - 1] < need) __builtin_amdgcn_s_sleep(2);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
                const unsigned long long t1 = k.dbg ? clock64() : 0;
                intra_mb_wave(k, L, wave, lane, s, x, row, t4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (k.dbg) { const unsigned long long t2 = clock64(); t_wait += t1 - t0; t_work += t2 - t1; n_mb++; }
                if (lane == 0) progress[row] = todo ? x0 + __builtin_ctzll(todo) : chunk_end;
            }
        }
    }
    if (k.dbg && lane == 0) {
        unsigned long long *d = k.dbg + ((size_t)s * 16 + wave) * 16;
        d[0] = t_wait; d[1] = t_work; d[2] = n_mb; d[3] = clock64() - t_begin;
        for (int i = 0; i < 6; i++) d[8 + i] = L.sect[wave][i];
    }
}

}  // namespace x264gpu

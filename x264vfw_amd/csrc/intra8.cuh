// intra8.cuh — Intra_8x8 prediction for gfx950 (A5: predict_8x8_filter + predict_8x8[9]; normative 8.3.2).
// Table-driven like the 4x4 predictors: the 25 reference samples of a block (l7..l0, tl, t0..t15) are low-pass
// filtered once (8.3.2.2.1), then every predicted pixel of every mode is one entry of a small per-block array
//   U8[k]      = E[k]                         k = 0..24   (E = filtered line)
//   U8[25 + k] = (E[k] + E[k+1] + 1) >> 1
//   U8[49 + k] = (E[k-1] + 2 E[k] + E[k+1] + 2) >> 2      (ends replicate)
//   U8[74]     = DC (left/top/128 variants chosen by availability)
// and c_pred8_table[mode][row][col] names the entry (generated from the CPU oracle by tools/gen_pred8_table.py).
// Restates oracle/predict.c x264o_predict_8x8_filter / x264o_predict_8x8 bit-exactly.
#pragma once
#include "intra.cuh"
#include "dsp8.cuh"

namespace x264gpu {

enum { U8_E = 0, U8_F2 = 25, U8_F3 = 49, U8_DC = 74, U8_SIZE = 80 };

static __constant__ __attribute__((aligned(8))) uint8_t c_pred8_table[9 * 64] = {
#include "pred8_table.inc"
};

// Build U8[] for the 8x8 block whose top-left sample is `blk` inside an LDS tile of stride ts whose row -1 /
// column -1 hold the neighbours.  All 64 lanes call; lane k < 25 owns line sample k.
__device__ __forceinline__ void pred8_build_u(uint8_t *U, const uint8_t *blk, int ts, int avail, int lane)
{
    int r = 0;
    if (lane < 8) r = blk[(7 - lane) * ts - 1];                          // left column, bottom -> top
    else if (lane == 8) r = blk[-ts - 1];                                // corner
    else if (lane < 25) {
        int x = lane - 9;
        if (x > 7 && !(avail & AVAIL_TOPRIGHT)) x = 7;                   // replicate top[7]
        r = blk[-ts + x];
    }
    const bool has_tl = avail & AVAIL_TOPLEFT;
    int a = __shfl_up(r, 1), b = __shfl_down(r, 1);
    if (lane == 0 || (lane == 9 && !has_tl)) a = r;                      // line ends / missing corner: 3:1 filters
    if (lane == 24 || (lane == 7 && !has_tl)) b = r;
    const int e = (a + 2 * r + b + 2) >> 2;
    int ea = __shfl_up(e, 1), eb = __shfl_down(e, 1);
    if (lane == 0) ea = e;
    if (lane == 24) eb = e;
    if (lane < 25) { U[U8_E + lane] = (uint8_t)e; U[U8_F3 + lane] = (uint8_t)((ea + 2 * e + eb + 2) >> 2); }
    if (lane < 24) U[U8_F2 + lane] = (uint8_t)((e + eb + 1) >> 1);
    const int sl = wave_sum(lane < 8 ? e : 0), st = wave_sum(lane >= 9 && lane < 17 ? e : 0);
    const bool l = avail & AVAIL_LEFT, t = avail & AVAIL_TOP;
    if (lane == 0) U[U8_DC] = (uint8_t)(l && t ? (st + sl + 8) >> 4 : l ? (sl + 4) >> 3 : t ? (st + 4) >> 3 : 128);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// one predicted row (8 pixels) of `mode`: tab = the predictor table (LDS copy), 8 indices per (mode,row)
__device__ __forceinline__ void pred8_row8(const uint8_t *U, const uint8_t *tab, int mode, int row, uint32_t &lo, uint32_t &hi)
{
    const uint2 t = *(const uint2 *)(tab + (mode * 8 + row) * 8);
    lo = (uint32_t)U[t.x & 0xff] | ((uint32_t)U[(t.x >> 8) & 0xff] << 8) | ((uint32_t)U[(t.x >> 16) & 0xff] << 16) | ((uint32_t)U[t.x >> 24] << 24);
    hi = (uint32_t)U[t.y & 0xff] | ((uint32_t)U[(t.y >> 8) & 0xff] << 8) | ((uint32_t)U[(t.y >> 16) & 0xff] << 16) | ((uint32_t)U[t.y >> 24] << 24);
}

}  // namespace x264gpu

// deblock_line.hip.h — the edge filter of 8.7.2 for one line of luma samples and its tables (shared by the picture-level loop filter, k_deblock.hip.h, and
// the macroblock loop's deblock-aware RD, k_mb.hip.h: x264_macroblock_deblock)
#pragma once
#include "enc_common.hip.h"

namespace x264gpu {

static __device__ const uint8_t d_alpha_table[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
    15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
static __device__ const uint8_t d_beta_table[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
static __device__ const uint8_t d_tc0_table[52][3] = {
    { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
    { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
    { 0, 0, 0 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 }, { 0, 1, 1 }, { 1, 1, 1 },
    { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 2, 3 },
    { 1, 2, 3 }, { 2, 2, 3 }, { 2, 2, 4 }, { 2, 3, 4 }, { 2, 3, 4 }, { 3, 3, 5 }, { 3, 4, 6 }, { 3, 4, 6 },
    { 4, 5, 7 }, { 4, 5, 8 }, { 4, 6, 9 }, { 5, 7, 10 }, { 6, 8, 11 }, { 6, 8, 13 }, { 7, 10, 14 }, { 8, 11, 16 },
    { 9, 12, 18 }, { 10, 13, 20 }, { 11, 15, 23 }, { 13, 17, 25 } };


// one luma line across an edge; pix -> q0, xs = byte step across the edge (oracle x264o_deblock_luma_edge)
__device__ __forceinline__ void filter_luma_line(uint8_t *pix, int xs, int alpha, int beta, int tc0, int bs)
{
    const int p2 = pix[-3 * xs], p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs];
    if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) return;
    const int ap = abs(p2 - p0), aq = abs(q2 - q0);
    if (bs < 4) {
        const int tc = tc0 + (ap < beta) + (aq < beta);
        const int delta = min(max((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc), tc);
        if (ap < beta) pix[-2 * xs] = (uint8_t)(p1 + min(max((p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1, -tc0), tc0));
        if (aq < beta) pix[xs] = (uint8_t)(q1 + min(max((q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1, -tc0), tc0));
        pix[-xs] = (uint8_t)clip_u8(p0 + delta);
        pix[0] = (uint8_t)clip_u8(q0 - delta);
    } else {
        const bool strong = abs(p0 - q0) < ((alpha >> 2) + 2);
        if (ap < beta && strong) {
            const int p3 = pix[-4 * xs];
            pix[-xs] = (uint8_t)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
            pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);
            pix[-3 * xs] = (uint8_t)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
        } else
            pix[-xs] = (uint8_t)((2 * p1 + p0 + q1 + 2) >> 2);
        if (aq < beta && strong) {
            const int q3 = pix[3 * xs];
            pix[0] = (uint8_t)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
            pix[xs] = (uint8_t)((p0 + q0 + q1 + q2 + 2) >> 2);
            pix[2 * xs] = (uint8_t)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
        } else
            pix[0] = (uint8_t)((2 * q1 + q0 + p1 + 2) >> 2);
    }
}


}  // namespace x264gpu

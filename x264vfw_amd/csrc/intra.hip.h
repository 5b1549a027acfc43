// intra.hip.h — H.264 intra predictors for wave64 (oracle/predict.c; normative 8.3.1-8.3.4).
//
// Neighbour samples live in small per-wave LDS arrays.  4x4 prediction is table driven: every
// directional predictor is a 2- or 3-tap filter over the "edge line" e[] (left column bottom->top,
// corner, top row left->right), so one LDS array U = { e, F2(e), F3(e), DC } plus a 9x16 index table
// yields all nine modes without divergence — one quad of lanes per mode, nine modes per wavefront.
#pragma once
#include "dsp.hip.h"

namespace x264gpu {

enum { AVAIL_LEFT = 1, AVAIL_TOP = 2, AVAIL_TOPRIGHT = 4, AVAIL_TOPLEFT = 8 };
enum { PRED16_V = 0, PRED16_H, PRED16_DC, PRED16_P, PRED16_DC_LEFT, PRED16_DC_TOP, PRED16_DC_128 };
enum { PREDC_DC = 0, PREDC_H, PREDC_V, PREDC_P, PREDC_DC_LEFT, PREDC_DC_TOP, PREDC_DC_128 };

// ---- luma neighbour array (per wave, 48 bytes): [0]=top-left, [1..24]=top row x=0..23, [32..47]=left
enum { NB_TL = 0, NB_TOP = 1, NB_LEFT = 32, NB_SIZE = 48 };

struct Pred16 {        // wave-uniform parameters of the 16x16 predictors
    int dc, dc_left, dc_top;
    int pa, pb, pc;    // plane: a, b, c
};

__device__ __forceinline__ Pred16 pred16_setup(const uint8_t *nb, int lane)
{
    Pred16 p;
    int t = lane < 16 ? nb[NB_TOP + lane] : 0, l = lane < 16 ? nb[NB_LEFT + lane] : 0;
    int st = wave_sum(t), sl = wave_sum(l);
    p.dc = (st + sl + 16) >> 5; p.dc_left = (sl + 8) >> 4; p.dc_top = (st + 8) >> 4;
    int h = 0, v = 0;
    if (lane < 8) {
        // top[-1] is the corner = nb[NB_TL] = nb[NB_TOP-1]; left[-1] likewise
        int lm = 6 - lane;
        h = (lane + 1) * (nb[NB_TOP + 8 + lane] - nb[NB_TOP + lm]);
        v = (lane + 1) * (nb[NB_LEFT + 8 + lane] - (lm >= 0 ? nb[NB_LEFT + lm] : nb[NB_TL]));
    }
    int H = wave_sum(h), V = wave_sum(v);
    p.pa = 16 * (nb[NB_LEFT + 15] + nb[NB_TOP + 15]);
    p.pb = (5 * H + 32) >> 6;
    p.pc = (5 * V + 32) >> 6;
    return p;
}

// 4 predicted luma pixels (x0..x0+3, y) of a 16x16 macroblock
__device__ __forceinline__ uint32_t pred16_row4(const uint8_t *nb, const Pred16 &p, int mode, int x0, int y)
{
    switch (mode) {
    case PRED16_V:
        return (uint32_t)nb[NB_TOP + x0] | ((uint32_t)nb[NB_TOP + x0 + 1] << 8) | ((uint32_t)nb[NB_TOP + x0 + 2] << 16) |
               ((uint32_t)nb[NB_TOP + x0 + 3] << 24);
    case PRED16_H: return 0x01010101u * nb[NB_LEFT + y];
    case PRED16_DC: return 0x01010101u * (uint32_t)p.dc;
    case PRED16_DC_LEFT: return 0x01010101u * (uint32_t)p.dc_left;
    case PRED16_DC_TOP: return 0x01010101u * (uint32_t)p.dc_top;
    case PRED16_DC_128: return 0x80808080u;
    default: {
        int base = p.pa + p.pb * (x0 - 7) + p.pc * (y - 7) + 16, v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = clip_u8((base + p.pb * i) >> 5);
        return pack4(v);
    }
    }
}

// ---- 4x4: table of indices into U[48] = { e[-1..13] (15), F2[0..12] (13), F3[0..12] (13), DC }
enum { U_E = 1 /* e[k] at U[1+k] */, U_F2 = 15, U_F3 = 28, U_DC = 41, U_SIZE = 48 };

struct Pred4Table { uint8_t t[9][16]; };
constexpr Pred4Table make_pred4_table()
{
    Pred4Table r{};
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) {
            int i = y * 4 + x;
            r.t[0][i] = (uint8_t)(U_E + 5 + x);           // V
            r.t[1][i] = (uint8_t)(U_E + 3 - y);           // H
            r.t[2][i] = (uint8_t)U_DC;                    // DC
            r.t[3][i] = (uint8_t)(U_F3 + 6 + x + y);      // DDL
            r.t[4][i] = (uint8_t)(U_F3 + 4 + x - y);      // DDR
            { int z = 2 * x - y, k = x - (y >> 1);        // VR
              r.t[5][i] = (uint8_t)(z >= 0 && !(z & 1) ? U_F2 + 4 + k : z > 0 ? U_F3 + 4 + k : z == -1 ? U_F3 + 4 : U_F3 + 5 - y + 2 * x); }
            { int z = 2 * y - x, k = y - (x >> 1);        // HD
              r.t[6][i] = (uint8_t)(z >= 0 && !(z & 1) ? U_F2 + 3 - k : z > 0 ? U_F3 + 4 - k : z == -1 ? U_F3 + 4 : U_F3 + 3 + x - 2 * y); }
            { int k = x + (y >> 1);                       // VL
              r.t[7][i] = (uint8_t)((y & 1) ? U_F3 + 6 + k : U_F2 + 5 + k); }
            { int z = x + 2 * y, k = y + (x >> 1);        // HU
              r.t[8][i] = (uint8_t)(z > 5 ? U_E + 0 : z == 5 ? U_F3 + 0 : (z & 1) ? U_F3 + 2 - k : U_F2 + 2 - k); }
        }
    return r;
}
static __constant__ Pred4Table c_pred4_table = make_pred4_table();

// which of the nine 4x4 modes may be used given the block's neighbour availability
__device__ __forceinline__ bool pred4_mode_ok(int mode, int avail)
{
    const bool l = avail & AVAIL_LEFT, t = avail & AVAIL_TOP, tl = avail & AVAIL_TOPLEFT;
    switch (mode) {
    case 0: case 3: case 7: return t;
    case 1: case 8: return l;
    case 2: return true;
    case 4: case 5: case 6: return l && t && tl;
    default: return false;
    }
}

// Build U[] for the 4x4 block whose top-left sample is `blk` inside an LDS tile of stride ts whose row -1 /
// column -1 hold the neighbours.  All 64 lanes call.  Lane k (0..14) fetches e[k-1]; the 2- and 3-tap
// filters come from DPP row shifts of that register (no LDS round trip); lane 15 computes the DC value.
__device__ __forceinline__ void pred4_build_u(uint8_t *U, const uint8_t *blk, int ts, int avail, int lane)
{
    int v = 0;
    if (lane < 15) {
        const int k = lane - 1;
        const int kk = k < 0 ? 0 : k > 12 ? 12 : k;
        if (kk <= 3) v = blk[(3 - kk) * ts - 1];                       // left column, bottom -> top
        else if (kk == 4) v = blk[-ts - 1];                           // corner
        else {
            int x = kk - 5;
            if (x > 3 && !(avail & AVAIL_TOPRIGHT)) x = 3;            // replicate top[3] (8.3.1.2)
            v = blk[-ts + x];
        }
    } else if (lane == 15) {
        const uint32_t t = *(const uint32_t *)(blk - ts);
        const int st = (int)__builtin_amdgcn_sad_u8(t, 0u, 0u);
        const int sl = blk[-1] + blk[ts - 1] + blk[2 * ts - 1] + blk[3 * ts - 1];
        const bool l = avail & AVAIL_LEFT, tp = avail & AVAIL_TOP;
        v = l && tp ? (st + sl + 4) >> 3 : l ? (sl + 2) >> 2 : tp ? (st + 2) >> 2 : 128;
    }
    const int lo = dpp<0x111>(v);     // row_shr:1  -> value of lane-1  (e[k-1])
    const int hi = dpp<0x101>(v);     // row_shl:1  -> value of lane+1  (e[k+1])
    if (lane < 15) U[lane] = (uint8_t)v;
    else if (lane == 15) U[U_DC] = (uint8_t)v;
    if (lane >= 1 && lane <= 13) {
        U[U_F2 + lane - 1] = (uint8_t)((v + hi + 1) >> 1);
        U[U_F3 + lane - 1] = (uint8_t)((lo + 2 * v + hi + 2) >> 2);
    }
    lds_order();
}

// predicted row (4 pixels) from U[] given this lane's four table indices packed in t4
__device__ __forceinline__ uint32_t pred4_row4(const uint8_t *U, uint32_t t4)
{
    return (uint32_t)U[t4 & 0xff] | ((uint32_t)U[(t4 >> 8) & 0xff] << 8) | ((uint32_t)U[(t4 >> 16) & 0xff] << 16) |
           ((uint32_t)U[t4 >> 24] << 24);
}

// ---- chroma 8x8 (per plane neighbour array of 17 bytes: [0]=tl, [1..8]=top, [9..16]=left)
enum { CNB_TL = 0, CNB_TOP = 1, CNB_LEFT = 9, CNB_SIZE = 20 };

struct PredC { int s0, s1, s2, s3, pa, pb, pc; };

// per-plane setup; every lane of the plane's 16-lane row gets the same values (nb = that plane's array)
__device__ __forceinline__ PredC predc_setup(const uint8_t *nb)
{
    PredC p;
    p.s0 = nb[CNB_TOP] + nb[CNB_TOP + 1] + nb[CNB_TOP + 2] + nb[CNB_TOP + 3];
    p.s1 = nb[CNB_TOP + 4] + nb[CNB_TOP + 5] + nb[CNB_TOP + 6] + nb[CNB_TOP + 7];
    p.s2 = nb[CNB_LEFT] + nb[CNB_LEFT + 1] + nb[CNB_LEFT + 2] + nb[CNB_LEFT + 3];
    p.s3 = nb[CNB_LEFT + 4] + nb[CNB_LEFT + 5] + nb[CNB_LEFT + 6] + nb[CNB_LEFT + 7];
    int H = 0, V = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int lo = 2 - i;   // index -1 is the corner
        H += (i + 1) * (nb[CNB_TOP + 4 + i] - (lo >= 0 ? nb[CNB_TOP + lo] : nb[CNB_TL]));
        V += (i + 1) * (nb[CNB_LEFT + 4 + i] - (lo >= 0 ? nb[CNB_LEFT + lo] : nb[CNB_TL]));
    }
    p.pa = 16 * (nb[CNB_LEFT + 7] + nb[CNB_TOP + 7]);
    p.pb = (34 * H + 32) >> 6;
    p.pc = (34 * V + 32) >> 6;
    return p;
}

// 4 predicted chroma pixels of 4x4 block i (raster in the 8x8), row j
__device__ __forceinline__ uint32_t predc_row4(const uint8_t *nb, const PredC p, int mode, int i, int j)
{
    const int x0 = (i & 1) * 4, y = (i >> 1) * 4 + j;
    switch (mode) {
    case PREDC_V:
        return (uint32_t)nb[CNB_TOP + x0] | ((uint32_t)nb[CNB_TOP + x0 + 1] << 8) | ((uint32_t)nb[CNB_TOP + x0 + 2] << 16) |
               ((uint32_t)nb[CNB_TOP + x0 + 3] << 24);
    case PREDC_H: return 0x01010101u * nb[CNB_LEFT + y];
    case PREDC_DC_128: return 0x80808080u;
    case PREDC_DC: {
        int dc = i == 0 ? (p.s0 + p.s2 + 4) >> 3 : i == 1 ? (p.s1 + 2) >> 2 : i == 2 ? (p.s3 + 2) >> 2 : (p.s1 + p.s3 + 4) >> 3;
        return 0x01010101u * (uint32_t)dc;
    }
    case PREDC_DC_LEFT: return 0x01010101u * (uint32_t)(((i >> 1) ? p.s3 + 2 : p.s2 + 2) >> 2);
    case PREDC_DC_TOP: return 0x01010101u * (uint32_t)(((i & 1) ? p.s1 + 2 : p.s0 + 2) >> 2);
    default: {
        int base = p.pa + p.pb * (x0 - 3) + p.pc * (y - 3) + 16, v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = clip_u8((base + p.pb * k) >> 5);
        return pack4(v);
    }
    }
}

}  // namespace x264gpu

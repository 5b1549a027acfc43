// oracle/cabac_dec.hpp — CABAC parsing process (ITU-T H.264 9.3) for the checker decoder (TEST INFRASTRUCTURE ONLY; see x264o.h).
//
// Written separately from the product's encoder (x264vfw_amd/host/cabac.cpp): its own arithmetic decoding engine (9.3.3.2) and its own
// copy of the context initialisation values, typed a second time and laid out per syntax element (Tables 9-12 .. 9-23) instead of per
// context index; tests/test_host_cpu.py checks the two copies against each other.  Like the encoder's copy they were typed from memory
// of the published tables — the standard's text is not in this container — so the pair pins typing errors, not recall errors.
// Supported: I, P and B slices, frame macroblocks, 4:2:0, cabac_init_idc 0.
#pragma once
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace cabacdec {

struct MN { int8_t m, n; };

// ---- rangeTabLPS (Table 9-44) and the state transitions (Table 9-45) ----
static const uint8_t kRangeLps[64][4] = {
    { 128, 176, 208, 240 }, { 128, 167, 197, 227 }, { 128, 158, 187, 216 }, { 123, 150, 178, 205 }, { 116, 142, 169, 195 }, { 111, 135, 160, 185 }, { 105, 128, 152, 175 },
    { 100, 122, 144, 166 }, { 95, 116, 137, 158 }, { 90, 110, 130, 150 }, { 85, 104, 123, 142 }, { 81, 99, 117, 135 }, { 77, 94, 111, 128 }, { 73, 89, 105, 122 },
    { 69, 85, 100, 116 }, { 66, 80, 95, 110 }, { 62, 76, 90, 104 }, { 59, 72, 86, 99 }, { 56, 69, 81, 94 }, { 53, 65, 77, 89 }, { 51, 62, 73, 85 },
    { 48, 59, 69, 80 }, { 46, 56, 66, 76 }, { 43, 53, 63, 72 }, { 41, 50, 59, 69 }, { 39, 48, 56, 65 }, { 37, 45, 54, 62 }, { 35, 43, 51, 59 },
    { 33, 41, 48, 56 }, { 32, 39, 46, 53 }, { 30, 37, 43, 50 }, { 29, 35, 41, 48 }, { 27, 33, 39, 45 }, { 26, 31, 37, 43 }, { 24, 30, 35, 41 },
    { 23, 28, 33, 39 }, { 22, 27, 32, 37 }, { 21, 26, 30, 35 }, { 20, 24, 29, 33 }, { 19, 23, 27, 31 }, { 18, 22, 26, 30 }, { 17, 21, 25, 28 },
    { 16, 20, 23, 27 }, { 15, 19, 22, 25 }, { 14, 18, 21, 24 }, { 14, 17, 20, 23 }, { 13, 16, 19, 22 }, { 12, 15, 18, 21 }, { 12, 14, 17, 20 },
    { 11, 14, 16, 19 }, { 11, 13, 15, 18 }, { 10, 12, 15, 17 }, { 10, 12, 14, 16 }, { 9, 11, 13, 15 }, { 9, 11, 12, 14 }, { 8, 10, 12, 14 },
    { 8, 9, 11, 13 }, { 7, 9, 11, 12 }, { 7, 9, 10, 12 }, { 7, 8, 10, 11 }, { 6, 8, 9, 11 }, { 6, 7, 9, 10 }, { 6, 7, 8, 9 }, { 2, 2, 2, 2 } };
static const uint8_t kNextLps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                      24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

// ---- context initialisation, per syntax element: [0] = I slices, [1] = P slices with cabac_init_idc 0 ----
// mb_type of I slices: ctxIdx 3..10 (0..2 belong to SI slices); the same values serve both columns
static const MN kMbTypeI[8] = { { 20, -15 }, { 2, 54 }, { 3, 74 }, { -28, 127 }, { -23, 104 }, { -6, 53 }, { -1, 54 }, { 7, 51 } };
// P slices: mb_skip_flag 11..13, mb_type 14..20, sub_mb_type 21..23
static const MN kSkipP[3] = { { 23, 33 }, { 23, 2 }, { 21, 0 } };
static const MN kMbTypeP[7] = { { 1, 9 }, { 0, 49 }, { -37, 118 }, { 5, 57 }, { -13, 78 }, { -11, 65 }, { 1, 62 } };
static const MN kSubMbTypeP[3] = { { 12, 49 }, { -4, 73 }, { 17, 50 } };
// B slices (cabac_init_idc 0): mb_skip_flag 24..26, mb_type 27..35, sub_mb_type 36..39
static const MN kSkipB[3] = { { 18, 64 }, { 9, 43 }, { 29, 0 } };
static const MN kMbTypeB[9] = { { 26, 67 }, { 16, 90 }, { 9, 104 }, { -46, 127 }, { -20, 104 }, { 1, 67 }, { -13, 78 }, { -11, 65 }, { 1, 62 } };
static const MN kSubMbTypeB[4] = { { -6, 86 }, { -17, 95 }, { -6, 61 }, { 9, 45 } };
// mvd_l0: horizontal 40..46, vertical 47..53; ref_idx 54..59
static const MN kMvdX[7] = { { -3, 69 }, { -6, 81 }, { -11, 96 }, { 6, 55 }, { 7, 67 }, { -5, 86 }, { 2, 88 } };
static const MN kMvdY[7] = { { 0, 58 }, { -3, 76 }, { -10, 94 }, { 5, 54 }, { 4, 69 }, { -3, 81 }, { 0, 88 } };
static const MN kRefIdx[6] = { { -7, 67 }, { -5, 74 }, { -4, 74 }, { -5, 80 }, { -7, 72 }, { 1, 58 } };
// mb_qp_delta 60..63, intra_chroma_pred_mode 64..67, prev_intra4x4_pred_mode_flag 68, rem_intra4x4_pred_mode 69 (all slice types)
static const MN kQpDelta[4] = { { 0, 41 }, { 0, 63 }, { 0, 63 }, { 0, 63 } };
static const MN kChromaPred[4] = { { -9, 83 }, { 4, 86 }, { 0, 97 }, { -7, 72 } };
static const MN kIntra4x4[2] = { { 13, 41 }, { 3, 62 } };
// coded_block_pattern: luma 73..76, chroma 77..84
static const MN kCbpLuma[2][4] = { { { -17, 127 }, { -13, 102 }, { 0, 82 }, { -7, 74 } }, { { -27, 126 }, { -28, 98 }, { -25, 101 }, { -23, 67 } } };
static const MN kCbpChroma[2][8] = { { { -21, 107 }, { -27, 127 }, { -31, 127 }, { -24, 127 }, { -18, 95 }, { -27, 127 }, { -21, 114 }, { -30, 127 } },
                                     { { -28, 82 }, { -20, 94 }, { -16, 83 }, { -22, 110 }, { -21, 91 }, { -18, 102 }, { -13, 93 }, { -29, 127 } } };
// coded_block_flag 85..104: block categories 0..4 x ctxIdxInc 0..3
static const MN kCbf[2][20] = {
    { { -17, 123 }, { -12, 115 }, { -16, 122 }, { -11, 115 }, { -12, 63 }, { -2, 68 }, { -15, 84 }, { -13, 104 }, { -3, 70 }, { -8, 93 }, { -10, 90 }, { -30, 127 },
      { -1, 74 }, { -6, 97 }, { -7, 91 }, { -20, 127 }, { -4, 56 }, { -5, 82 }, { -7, 76 }, { -22, 125 } },
    { { -7, 92 }, { -5, 89 }, { -7, 96 }, { -13, 108 }, { -3, 46 }, { -1, 65 }, { -1, 57 }, { -9, 93 }, { -3, 74 }, { -9, 92 }, { -8, 87 }, { -23, 126 },
      { 5, 54 }, { 6, 60 }, { 6, 59 }, { 6, 69 }, { -1, 48 }, { 0, 68 }, { -4, 69 }, { -8, 88 } } };
// significant_coeff_flag, frame macroblocks, 105..165: categories 0 (15), 1 (14), 2 (15), 3 (3), 4 (14)
static const MN kSig[2][61] = {
    { { -7, 93 }, { -11, 87 }, { -3, 77 }, { -5, 71 }, { -4, 63 }, { -4, 68 }, { -12, 84 }, { -7, 62 }, { -7, 65 }, { 8, 61 }, { 5, 56 }, { -2, 66 }, { 1, 64 }, { 0, 61 }, { -2, 78 },
      { 1, 50 }, { 7, 52 }, { 10, 35 }, { 0, 44 }, { 11, 38 }, { 1, 45 }, { 0, 46 }, { 5, 44 }, { 31, 17 }, { 1, 51 }, { 7, 50 }, { 28, 19 }, { 16, 33 }, { 14, 62 },
      { -13, 108 }, { -15, 100 }, { -13, 101 }, { -13, 91 }, { -12, 94 }, { -10, 88 }, { -16, 84 }, { -10, 86 }, { -7, 83 }, { -13, 87 }, { -19, 94 }, { 1, 70 }, { 0, 72 }, { -5, 74 }, { 18, 59 },
      { -8, 102 }, { -15, 100 }, { 0, 95 },
      { -4, 75 }, { 2, 72 }, { -11, 75 }, { -3, 71 }, { 15, 46 }, { -13, 69 }, { 0, 62 }, { 0, 65 }, { 21, 37 }, { -15, 72 }, { 9, 57 }, { 16, 54 }, { 0, 62 }, { 12, 72 } },
    { { -2, 85 }, { -6, 78 }, { -1, 75 }, { -7, 77 }, { 2, 54 }, { 5, 50 }, { -3, 68 }, { 1, 50 }, { 6, 42 }, { -4, 81 }, { 1, 63 }, { -4, 70 }, { 0, 67 }, { 2, 57 }, { -2, 76 },
      { 11, 35 }, { 4, 64 }, { 1, 61 }, { 11, 35 }, { 18, 25 }, { 12, 24 }, { 13, 29 }, { 13, 36 }, { -10, 93 }, { -7, 73 }, { -2, 73 }, { 13, 46 }, { 9, 49 }, { -7, 100 },
      { 9, 53 }, { 2, 53 }, { 5, 53 }, { -2, 61 }, { 0, 56 }, { 0, 56 }, { -13, 63 }, { -5, 60 }, { -1, 62 }, { 4, 57 }, { -6, 69 }, { 4, 57 }, { 14, 39 }, { 4, 51 }, { 13, 68 },
      { 3, 64 }, { 1, 61 }, { 9, 63 },
      { 7, 50 }, { 16, 39 }, { 5, 44 }, { 4, 52 }, { 11, 48 }, { -5, 60 }, { -1, 59 }, { 0, 59 }, { 22, 33 }, { 5, 44 }, { 14, 43 }, { -1, 78 }, { 0, 60 }, { 9, 69 } } };
// last_significant_coeff_flag, frame macroblocks, 166..226 (same category split)
static const MN kLast[2][61] = {
    { { 24, 0 }, { 15, 9 }, { 8, 25 }, { 13, 18 }, { 15, 9 }, { 13, 19 }, { 10, 37 }, { 12, 18 }, { 6, 29 }, { 20, 33 }, { 15, 30 }, { 4, 45 }, { 1, 58 }, { 0, 62 }, { 7, 61 },
      { 12, 38 }, { 11, 45 }, { 15, 39 }, { 11, 42 }, { 13, 44 }, { 16, 45 }, { 12, 41 }, { 10, 49 }, { 30, 34 }, { 18, 42 }, { 10, 55 }, { 17, 51 }, { 17, 46 }, { 0, 89 },
      { 26, -19 }, { 22, -17 }, { 26, -17 }, { 30, -25 }, { 28, -20 }, { 33, -23 }, { 37, -27 }, { 33, -23 }, { 40, -28 }, { 38, -17 }, { 33, -11 }, { 40, -15 }, { 41, -6 }, { 38, 1 }, { 41, 17 },
      { 30, -6 }, { 27, 3 }, { 26, 22 },
      { 37, -16 }, { 35, -4 }, { 38, -8 }, { 38, -3 }, { 37, 3 }, { 38, 5 }, { 42, 0 }, { 35, 16 }, { 39, 22 }, { 14, 48 }, { 27, 37 }, { 21, 60 }, { 12, 68 }, { 2, 97 } },
    { { 11, 28 }, { 2, 40 }, { 3, 44 }, { 0, 49 }, { 0, 46 }, { 2, 44 }, { 2, 51 }, { 0, 47 }, { 4, 39 }, { 2, 62 }, { 6, 46 }, { 0, 54 }, { 3, 54 }, { 2, 58 }, { 4, 63 },
      { 6, 51 }, { 6, 57 }, { 7, 53 }, { 6, 52 }, { 6, 55 }, { 11, 45 }, { 14, 36 }, { 8, 53 }, { -1, 82 }, { 7, 55 }, { -3, 78 }, { 15, 46 }, { 22, 31 }, { -1, 84 },
      { 25, 7 }, { 30, -7 }, { 28, 3 }, { 28, 4 }, { 32, 0 }, { 34, -1 }, { 30, 6 }, { 30, 6 }, { 32, 9 }, { 31, 19 }, { 26, 27 }, { 26, 30 }, { 37, 20 }, { 28, 34 }, { 17, 70 },
      { 1, 67 }, { 5, 59 }, { 9, 67 },
      { 16, 30 }, { 18, 32 }, { 18, 35 }, { 22, 29 }, { 24, 31 }, { 23, 38 }, { 18, 43 }, { 20, 41 }, { 11, 63 }, { 9, 59 }, { 9, 64 }, { -1, 94 }, { -2, 89 }, { -9, 108 } } };
// coeff_abs_level_minus1 227..275: categories 0..2 ten each, 3 nine, 4 ten
static const MN kAbs[2][49] = {
    { { -3, 71 }, { -6, 42 }, { -5, 50 }, { -3, 54 }, { -2, 62 }, { 0, 58 }, { 1, 63 }, { -2, 72 }, { -1, 74 }, { -9, 91 },
      { -5, 67 }, { -5, 27 }, { -3, 39 }, { -2, 44 }, { 0, 46 }, { -16, 64 }, { -8, 68 }, { -10, 78 }, { -6, 77 }, { -10, 86 },
      { -12, 92 }, { -15, 55 }, { -10, 60 }, { -6, 62 }, { -4, 65 }, { -12, 73 }, { -8, 76 }, { -7, 80 }, { -9, 88 }, { -17, 110 },
      { -11, 97 }, { -20, 84 }, { -11, 79 }, { -6, 73 }, { -4, 74 }, { -13, 86 }, { -13, 96 }, { -11, 97 }, { -19, 117 },
      { -8, 78 }, { -5, 33 }, { -4, 48 }, { -2, 53 }, { -3, 62 }, { -13, 71 }, { -10, 79 }, { -12, 86 }, { -13, 90 }, { -14, 97 } },
    { { -6, 76 }, { -2, 44 }, { 0, 45 }, { 0, 52 }, { -3, 64 }, { -2, 59 }, { -4, 70 }, { -4, 75 }, { -8, 82 }, { -17, 102 },
      { -9, 77 }, { 3, 24 }, { 0, 42 }, { 0, 48 }, { 0, 55 }, { -6, 59 }, { -7, 71 }, { -12, 83 }, { -11, 87 }, { -30, 119 },
      { 1, 58 }, { -3, 29 }, { -1, 36 }, { 1, 38 }, { 2, 43 }, { -6, 55 }, { 0, 58 }, { 0, 64 }, { -3, 74 }, { -10, 90 },
      { 0, 70 }, { -4, 29 }, { 5, 31 }, { 7, 42 }, { 1, 59 }, { -2, 58 }, { -3, 72 }, { -3, 81 }, { -11, 97 },
      { 0, 58 }, { 8, 5 }, { 10, 14 }, { 14, 18 }, { 13, 27 }, { 2, 40 }, { 0, 58 }, { -3, 70 }, { -6, 79 }, { -8, 85 } } };
// transform_size_8x8_flag 399..401; 8x8 blocks: significant_coeff_flag 402..416, last_significant_coeff_flag 417..425, coeff_abs_level_minus1 426..435
static const MN kT8[2][3] = { { { 31, 21 }, { 31, 31 }, { 25, 50 } }, { { 12, 40 }, { 11, 51 }, { 14, 59 } } };
static const MN kSig8[2][15] = {
    { { -17, 120 }, { -20, 112 }, { -18, 114 }, { -11, 85 }, { -15, 92 }, { -14, 89 }, { -26, 71 }, { -15, 81 }, { -14, 80 }, { 0, 68 }, { -14, 70 }, { -24, 56 }, { -23, 68 }, { -24, 50 }, { -11, 74 } },
    { { -4, 79 }, { -7, 71 }, { -5, 69 }, { -9, 70 }, { -8, 66 }, { -10, 68 }, { -19, 73 }, { -12, 69 }, { -16, 70 }, { -15, 67 }, { -20, 62 }, { -19, 70 }, { -16, 66 }, { -22, 65 }, { -20, 63 } } };
static const MN kLast8[2][9] = { { { 23, -13 }, { 26, -13 }, { 40, -15 }, { 49, -14 }, { 44, 3 }, { 45, 6 }, { 44, 34 }, { 33, 54 }, { 19, 82 } },
                                 { { 9, -2 }, { 26, -9 }, { 33, -9 }, { 39, -7 }, { 41, -2 }, { 45, 3 }, { 49, 9 }, { 45, 27 }, { 36, 59 } } };
static const MN kAbs8[2][10] = { { { -3, 75 }, { -1, 23 }, { 1, 34 }, { 1, 43 }, { 0, 54 }, { -2, 55 }, { 0, 61 }, { 1, 64 }, { 0, 68 }, { -9, 92 } },
                                 { { -6, 66 }, { -7, 35 }, { -7, 42 }, { -8, 45 }, { -5, 48 }, { -12, 56 }, { -6, 60 }, { -5, 62 }, { -8, 66 }, { -8, 76 } } };
// position -> ctxIdxInc of the 8x8 significance maps, frame macroblocks (Table 9-43)
static const uint8_t kSigInc8[63] = { 0, 1, 2, 3, 4, 5, 5, 4, 4, 3, 3, 4, 4, 4, 5, 5, 4, 4, 4, 4, 3, 3, 6, 7, 7, 7, 8, 9, 10, 9, 8, 7, 7, 6, 11, 12, 13, 11, 6, 7, 8, 9, 14, 10, 9, 8, 6, 11,
                                      12, 13, 11, 6, 9, 14, 10, 9, 11, 12, 13, 11, 14, 10, 12 };
static const uint8_t kLastInc8[63] = { 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,
                                       5, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8 };

// one context variable per ctxIdx (0..459)
struct Engine {
    const uint8_t *p = nullptr; size_t nbytes = 0, pos = 0;      // pos in bits
    uint32_t range = 510, offset = 0;
    uint8_t st[460], mps[460];
    bool err = false;

    int bit() { if (pos >= nbytes * 8) { err = true; return 0; } const int b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
    void seed(int ctx, MN v, int qp)
    {
        int pre = ((v.m * qp) >> 4) + v.n;
        pre = pre < 1 ? 1 : pre > 126 ? 126 : pre;
        if (pre <= 63) { st[ctx] = (uint8_t)(63 - pre); mps[ctx] = 0; } else { st[ctx] = (uint8_t)(pre - 64); mps[ctx] = 1; }
    }
    // 9.3.1.1 + 9.3.1.2: context variables from SliceQPY, then the arithmetic decoding engine
    void start(const uint8_t *data, size_t n, size_t bitpos, bool pslice, int qp)
    {
        p = data; nbytes = n; pos = bitpos;
        qp = qp < 0 ? 0 : qp > 51 ? 51 : qp;
        memset(st, 0, sizeof(st)); memset(mps, 0, sizeof(mps));
        const int t = pslice ? 1 : 0;
        for (int i = 0; i < 8; i++) seed(3 + i, kMbTypeI[i], qp);
        if (pslice) {
            for (int i = 0; i < 3; i++) seed(11 + i, kSkipP[i], qp);
            for (int i = 0; i < 7; i++) seed(14 + i, kMbTypeP[i], qp);
            for (int i = 0; i < 3; i++) seed(21 + i, kSubMbTypeP[i], qp);
            for (int i = 0; i < 3; i++) seed(24 + i, kSkipB[i], qp);
            for (int i = 0; i < 9; i++) seed(27 + i, kMbTypeB[i], qp);
            for (int i = 0; i < 4; i++) seed(36 + i, kSubMbTypeB[i], qp);
            for (int i = 0; i < 7; i++) { seed(40 + i, kMvdX[i], qp); seed(47 + i, kMvdY[i], qp); }
            for (int i = 0; i < 6; i++) seed(54 + i, kRefIdx[i], qp);
        }
        for (int i = 0; i < 4; i++) { seed(60 + i, kQpDelta[i], qp); seed(64 + i, kChromaPred[i], qp); seed(73 + i, kCbpLuma[t][i], qp); }
        seed(68, kIntra4x4[0], qp); seed(69, kIntra4x4[1], qp);
        for (int i = 0; i < 8; i++) seed(77 + i, kCbpChroma[t][i], qp);
        for (int i = 0; i < 20; i++) seed(85 + i, kCbf[t][i], qp);
        for (int i = 0; i < 61; i++) { seed(105 + i, kSig[t][i], qp); seed(166 + i, kLast[t][i], qp); }
        for (int i = 0; i < 49; i++) seed(227 + i, kAbs[t][i], qp);
        for (int i = 0; i < 3; i++) seed(399 + i, kT8[t][i], qp);
        for (int i = 0; i < 15; i++) seed(402 + i, kSig8[t][i], qp);
        for (int i = 0; i < 9; i++) seed(417 + i, kLast8[t][i], qp);
        for (int i = 0; i < 10; i++) seed(426 + i, kAbs8[t][i], qp);
        range = 510; offset = 0;
        for (int i = 0; i < 9; i++) offset = (offset << 1) | (uint32_t)bit();
    }
    int nbins = 0;
    int decision(int ctx)
    {
        const int r_ = decision_(ctx);
        if (nbins++ < 24 && getenv("X264O_CABAC_DEBUG")) fprintf(stderr, "dec bin %d ctx %d -> %d\n", nbins - 1, ctx, r_);
        return r_;
    }
    int decision_(int ctx)
    {
        const int s = st[ctx];
        const uint32_t lps = kRangeLps[s][(range >> 6) & 3];
        int bin;
        range -= lps;
        if (offset >= range) {
            bin = !mps[ctx]; offset -= range; range = lps;
            if (!s) mps[ctx] ^= 1;
            st[ctx] = kNextLps[s];
        } else { bin = mps[ctx]; st[ctx] = (uint8_t)(s < 62 ? s + 1 : 62); }
        while (range < 256) { range <<= 1; offset = (offset << 1) | (uint32_t)bit(); }
        return bin;
    }
    int bypass()
    {
        offset = (offset << 1) | (uint32_t)bit();
        if (offset >= range) { offset -= range; return 1; }
        return 0;
    }
    int terminate()
    {
        range -= 2;
        if (offset >= range) return 1;
        while (range < 256) { range <<= 1; offset = (offset << 1) | (uint32_t)bit(); }
        return 0;
    }
    int golomb_bypass(int k)          // UEGk suffix
    {
        int v = 0;
        while (bypass()) { v += 1 << k; k++; if (k > 24) { err = true; return 0; } }
        while (k--) v += bypass() << k;
        return v;
    }
};

}  // namespace cabacdec

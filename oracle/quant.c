/* oracle/quant.c — quantisation / dequantisation (TEST INFRASTRUCTURE; see x264o.h header).
 * Dequant follows ITU-T H.264 8.5.9-8.5.13 (normative).  Forward quant restates
 * [x264-upstream] common/quant.c + common/set.c (deadzone quant, flat CQM), SURVEY.md Appendix C;
 * deadzone defaults 21/11 are the values the reference prints at config.c:1688-1689. */
#include "x264o.h"
#include <stdlib.h>
#include <string.h>

const uint8_t x264o_chroma_qp[52] = {
    0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
    18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
    34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };

/* normAdjust4x4 (8-315): class 0 = (even,even), 1 = mixed, 2 = (odd,odd) */
static const uint8_t dequant4_scale[6][3] = {
    { 10, 13, 16 }, { 11, 14, 18 }, { 13, 16, 20 }, { 14, 18, 23 }, { 16, 20, 25 }, { 18, 23, 29 } };
static const uint16_t quant4_scale[6][3] = {
    { 13107, 8066, 5243 }, { 11916, 7490, 4660 }, { 10082, 6554, 4194 },
    { 9362, 5825, 3647 },  { 8192, 5243, 3355 },  { 7282, 4559, 2893 } };
/* normAdjust8x8 (8-318) classes 0..5 */
static const uint8_t dequant8_scale[6][6] = {
    { 20, 18, 32, 19, 25, 24 }, { 22, 19, 35, 21, 28, 26 }, { 26, 23, 42, 24, 33, 31 },
    { 28, 25, 45, 26, 35, 33 }, { 32, 28, 51, 30, 40, 38 }, { 36, 32, 58, 34, 46, 43 } };
static const uint16_t quant8_scale[6][6] = {
    { 13107, 11428, 20972, 12222, 16777, 15481 }, { 11916, 10826, 19174, 11058, 14980, 14290 },
    { 10082, 8943, 15978, 9675, 12710, 11985 },   { 9362, 8228, 14913, 8931, 11984, 11259 },
    { 8192, 7346, 13159, 7740, 10486, 9777 },     { 7282, 6428, 11570, 6830, 9118, 8640 } };

static int class4(int i) { return (i & 1) + ((i >> 2) & 1); }
static int class8(int i)
{
    int r = (i >> 3) & 3, c = i & 3;
    static const uint8_t cls[4][4] = { { 0, 3, 4, 3 }, { 3, 1, 5, 1 }, { 4, 5, 2, 5 }, { 3, 1, 5, 1 } };
    return cls[r][c];
}

static int shift_round(int x, int s) { return s <= 0 ? x << -s : (x + (1 << (s - 1))) >> s; }
static int div_round(int n, int d) { return (n + (d >> 1)) / d; }

void x264o_quant_init(x264o_quant_tables *t, int deadzone_inter, int deadzone_intra)
{
    /* list order: intra-luma, inter-luma, intra-chroma, inter-chroma (chroma deadzones fixed 11/21) */
    const int dz4[4] = { 32 - deadzone_intra, 32 - deadzone_inter, 32 - 11, 32 - 21 };
    const int dz8[2] = { 32 - deadzone_intra, 32 - deadzone_inter };
    memset(t, 0, sizeof(*t));
    for (int q = 0; q < 6; q++) {
        for (int i = 0; i < 16; i++) t->dequant4_mf[q][i] = dequant4_scale[q][class4(i)] * 16;
        for (int i = 0; i < 64; i++) t->dequant8_mf[q][i] = dequant8_scale[q][class8(i)] * 16;
    }
    for (int qp = 0; qp < 52; qp++) {
        for (int l = 0; l < 4; l++)
            for (int i = 0; i < 16; i++) {
                int mf = shift_round(quant4_scale[qp % 6][class4(i)], qp / 6 - 1);
                t->quant4_mf[l][qp][i] = (uint16_t)mf;
                int b = div_round(dz4[l] << 10, mf), cap = (1 << 15) / mf;
                t->quant4_bias[l][qp][i] = (uint16_t)(b < cap ? b : cap);
            }
        for (int l = 0; l < 2; l++)
            for (int i = 0; i < 64; i++) {
                int mf = shift_round(quant8_scale[qp % 6][class8(i)], qp / 6);
                t->quant8_mf[l][qp][i] = (uint16_t)mf;
                int b = div_round(dz8[l] << 10, mf), cap = (1 << 15) / mf;
                t->quant8_bias[l][qp][i] = (uint16_t)(b < cap ? b : cap);
            }
    }
}

/* trellis: the direct inverse of the quantiser in the forward transform's scale (x264 unquant4_mf / unquant8_mf, flat matrices) */
int x264o_unquant4(int qp, int pos) { return (int)((1ull << (qp / 6 + 15 + 8)) / quant4_scale[qp % 6][class4(pos)]); }
int x264o_unquant8(int qp, int pos) { return (int)((1ull << (qp / 6 + 16 + 8)) / quant8_scale[qp % 6][class8(pos)]); }

static inline int quant_one(int c, int mf, int bias)
{
    return c > 0 ? ((bias + c) * mf) >> 16 : -(((bias - c) * mf) >> 16);
}

int x264o_quant_4x4(dctcoef d[16], const uint16_t mf[16], const uint16_t bias[16])
{
    int nz = 0;
    for (int i = 0; i < 16; i++) nz |= (d[i] = (dctcoef)quant_one(d[i], mf[i], bias[i]));
    return !!nz;
}

int x264o_quant_8x8(dctcoef d[64], const uint16_t mf[64], const uint16_t bias[64])
{
    int nz = 0;
    for (int i = 0; i < 64; i++) nz |= (d[i] = (dctcoef)quant_one(d[i], mf[i], bias[i]));
    return !!nz;
}

int x264o_quant_4x4_dc(dctcoef d[16], int mf, int bias)
{
    int nz = 0;
    for (int i = 0; i < 16; i++) nz |= (d[i] = (dctcoef)quant_one(d[i], mf, bias));
    return !!nz;
}

int x264o_quant_2x2_dc(dctcoef d[4], int mf, int bias)
{
    int nz = 0;
    for (int i = 0; i < 4; i++) nz |= (d[i] = (dctcoef)quant_one(d[i], mf, bias));
    return !!nz;
}

/* 8.5.12.1: residual 4x4 scaling, LevelScale4x4 = 16 * normAdjust (flat weights) */
void x264o_dequant_4x4(dctcoef d[16], const int32_t dq[6][16], int qp)
{
    int m = qp % 6, s = qp / 6 - 4;
    for (int i = 0; i < 16; i++)
        d[i] = (dctcoef)(s >= 0 ? (d[i] * dq[m][i]) * (1 << s) : (d[i] * dq[m][i] + (1 << (-s - 1))) >> -s);
}

/* 8.5.13 scaling for 8x8 blocks */
void x264o_dequant_8x8(dctcoef d[64], const int32_t dq[6][64], int qp)
{
    int m = qp % 6, s = qp / 6 - 6;
    for (int i = 0; i < 64; i++)
        d[i] = (dctcoef)(s >= 0 ? (d[i] * dq[m][i]) * (1 << s) : (d[i] * dq[m][i] + (1 << (-s - 1))) >> -s);
}

/* 8.5.10: Intra16x16 luma DC scaling (input already inverse-Hadamard transformed) */
void x264o_dequant_4x4_dc(dctcoef d[16], const int32_t dq[6][16], int qp)
{
    int ls = dq[qp % 6][0], s = qp / 6 - 6;
    for (int i = 0; i < 16; i++)
        d[i] = (dctcoef)(s >= 0 ? (d[i] * ls) * (1 << s) : (d[i] * ls + (1 << (-s - 1))) >> -s);
}

/* 8.5.11.2 (4:2:0): f = H c H, dcC = ((f * LevelScale(qp%6,0,0)) << (qp/6)) >> 5 */
void x264o_dequant_2x2_dc(dctcoef out[4], const dctcoef in[4], const int32_t dq[6][16], int qp)
{
    dctcoef f[4] = { in[0], in[1], in[2], in[3] };
    x264o_dct2x2dc(f);
    int ls = dq[qp % 6][0] << (qp / 6);
    for (int i = 0; i < 4; i++) out[i] = (dctcoef)((f[i] * ls) >> 5);
}

/* optimize_chroma_2x2_dc ([x264-upstream] common/quant.c, used by x264_mb_encode_chroma_internal on the DC-only path): lower the
 * magnitude of the quantised chroma DC levels as long as the reconstruction — ((idct2x2(level) * dmf) >> 5) + 32 >> 6 per 4x4
 * block — does not change; highest frequency first.  d[] in this file's layout (0 sum, 1 horizontal, 2 vertical, 3 diagonal;
 * x264 keeps 1 and 2 the other way round, hence the visiting order 3, 1, 2, 0).  Returns 0 when nothing remains to code. */
static void idct_dequant_round_2x2(int out[4], const dctcoef d[4], int dmf)
{
    int d0 = d[0] + d[1], d1 = d[2] + d[3], d2 = d[0] - d[1], d3 = d[2] - d[3];
    out[0] = ((d0 + d1) * dmf >> 5) + 32;
    out[1] = ((d0 - d1) * dmf >> 5) + 32;
    out[2] = ((d2 + d3) * dmf >> 5) + 32;
    out[3] = ((d2 - d3) * dmf >> 5) + 32;
}

int x264o_optimize_chroma_2x2_dc(dctcoef d[4], int dmf)
{
    static const int order[4] = { 3, 1, 2, 0 };
    int ref[4], out[4], sum = 0, nz = 0;
    if (dmf > 32 * 64) return 1;                 /* x264_mb_optimize_chroma_dc: large quantisers are left alone */
    idct_dequant_round_2x2(ref, d, dmf);
    for (int i = 0; i < 4; i++) sum |= ref[i];
    if (!(sum >> 6)) return 0;                   /* already rounds to nothing */
    for (int k = 0; k < 4; k++) {
        int c = order[k], level = d[c], sign = level >> 31 | 1;
        while (level) {
            d[c] = (dctcoef)(level - sign);
            idct_dequant_round_2x2(out, d, dmf);
            if (((ref[0] ^ out[0]) | (ref[1] ^ out[1]) | (ref[2] ^ out[2]) | (ref[3] ^ out[3])) >> 6) { nz = 1; d[c] = (dctcoef)level; break; }
            level -= sign;
        }
    }
    return nz;
}

int x264o_coeff_last(const dctcoef *l, int n)
{
    int i = n - 1;
    while (i >= 0 && l[i] == 0) i--;
    return i;
}

/* dct-decimate score over scan-ordered levels: 9 as soon as any |level| > 1, else a run-length
 * weighted count of the +-1 levels (small isolated coefficients are cheap to drop). */
int x264o_decimate_score(const dctcoef *l, int n)
{
    static const uint8_t t4[16] = { 3, 2, 2, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    static const uint8_t t8[64] = { 3, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1,
                                    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1 };
    const uint8_t *tab = n == 64 ? t8 : t4;
    int idx = x264o_coeff_last(l, n), score = 0;
    while (idx >= 0) {
        if (abs(l[idx--]) > 1) return 9;
        int run = 0;
        while (idx >= 0 && l[idx] == 0) { idx--; run++; }
        score += tab[run];
    }
    return score;
}

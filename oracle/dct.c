/* oracle/dct.c — H.264 integer transforms (TEST INFRASTRUCTURE; see x264o.h header).
 * Inverse transforms follow ITU-T H.264 8.5.10-8.5.13 (normative: rows first, then columns).
 * Forward transforms restate [x264-upstream] common/dct.c (the JM core transform; scaling is
 * folded into quant).  Reached in the reference only via x264_encoder_encode() (codec.c:1693). */
#include "x264o.h"

const uint8_t x264o_zigzag4[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
const uint8_t x264o_zigzag8[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

static inline pixel clip_pixel(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

static void fwd4_1d(const int *in, int si, int *out, int so)
{
    int s03 = in[0] + in[3 * si], s12 = in[si] + in[2 * si];
    int d03 = in[0] - in[3 * si], d12 = in[si] - in[2 * si];
    out[0] = s03 + s12;
    out[so] = 2 * d03 + d12;
    out[2 * so] = s03 - s12;
    out[3 * so] = d03 - 2 * d12;
}

void x264o_sub4x4_dct(dctcoef d[16], const pixel *enc, int se, const pixel *pred, int sp)
{
    int r[16], t[16], o[16];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) r[y * 4 + x] = enc[y * se + x] - pred[y * sp + x];
    for (int y = 0; y < 4; y++) fwd4_1d(r + y * 4, 1, t + y * 4, 1);
    for (int x = 0; x < 4; x++) fwd4_1d(t + x, 4, o + x, 4);
    for (int i = 0; i < 16; i++) d[i] = (dctcoef)o[i];
}

static void inv4_1d(const int *in, int si, int *out, int so)
{
    int e0 = in[0] + in[2 * si];
    int e1 = in[0] - in[2 * si];
    int e2 = (in[si] >> 1) - in[3 * si];
    int e3 = in[si] + (in[3 * si] >> 1);
    out[0] = e0 + e3;
    out[so] = e1 + e2;
    out[2 * so] = e1 - e2;
    out[3 * so] = e0 - e3;
}

void x264o_add4x4_idct(pixel *dst, int sd, const dctcoef d[16])
{
    int c[16], f[16], g[16];
    for (int i = 0; i < 16; i++) c[i] = d[i];
    for (int y = 0; y < 4; y++) inv4_1d(c + y * 4, 1, f + y * 4, 1); /* rows (8-338..8-345) */
    for (int x = 0; x < 4; x++) inv4_1d(f + x, 4, g + x, 4);          /* columns (8-346..8-353) */
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            dst[y * sd + x] = clip_pixel(dst[y * sd + x] + ((g[y * 4 + x] + 32) >> 6));
}

void x264o_add4x4_idct_dc(pixel *dst, int sd, int dc)
{
    dc = (dc + 32) >> 6;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++) dst[y * sd + x] = clip_pixel(dst[y * sd + x] + dc);
}

static void fwd8_1d(const int *s, int si, int *o, int so)
{
    int s07 = s[0] + s[7 * si], s16 = s[si] + s[6 * si], s25 = s[2 * si] + s[5 * si], s34 = s[3 * si] + s[4 * si];
    int d07 = s[0] - s[7 * si], d16 = s[si] - s[6 * si], d25 = s[2 * si] - s[5 * si], d34 = s[3 * si] - s[4 * si];
    int a0 = s07 + s34, a1 = s16 + s25, a2 = s07 - s34, a3 = s16 - s25;
    int a4 = d16 + d25 + (d07 + (d07 >> 1));
    int a5 = d07 - d34 - (d25 + (d25 >> 1));
    int a6 = d07 + d34 - (d16 + (d16 >> 1));
    int a7 = d16 - d25 + (d34 + (d34 >> 1));
    o[0] = a0 + a1;
    o[so] = a4 + (a7 >> 2);
    o[2 * so] = a2 + (a3 >> 1);
    o[3 * so] = a5 + (a6 >> 2);
    o[4 * so] = a0 - a1;
    o[5 * so] = a6 - (a5 >> 2);
    o[6 * so] = (a2 >> 1) - a3;
    o[7 * so] = (a4 >> 2) - a7;
}

void x264o_sub8x8_dct8(dctcoef d[64], const pixel *enc, int se, const pixel *pred, int sp)
{
    int r[64], t[64], o[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++) r[y * 8 + x] = enc[y * se + x] - pred[y * sp + x];
    for (int y = 0; y < 8; y++) fwd8_1d(r + y * 8, 1, t + y * 8, 1);
    for (int x = 0; x < 8; x++) fwd8_1d(t + x, 8, o + x, 8);
    for (int i = 0; i < 64; i++) d[i] = (dctcoef)o[i];
}

/* 8.5.13: one-dimensional 8-point inverse */
static void inv8_1d(const int *s, int si, int *o, int so)
{
    int a0 = s[0] + s[4 * si];
    int a2 = s[0] - s[4 * si];
    int a4 = (s[2 * si] >> 1) - s[6 * si];
    int a6 = (s[6 * si] >> 1) + s[2 * si];
    int b0 = a0 + a6, b2 = a2 + a4, b4 = a2 - a4, b6 = a0 - a6;
    int a1 = -s[3 * si] + s[5 * si] - s[7 * si] - (s[7 * si] >> 1);
    int a3 = s[si] + s[7 * si] - s[3 * si] - (s[3 * si] >> 1);
    int a5 = -s[si] + s[7 * si] + s[5 * si] + (s[5 * si] >> 1);
    int a7 = s[3 * si] + s[5 * si] + s[si] + (s[si] >> 1);
    int b1 = (a7 >> 2) + a1;
    int b3 = a3 + (a5 >> 2);
    int b5 = (a3 >> 2) - a5;
    int b7 = a7 - (a1 >> 2);
    o[0] = b0 + b7;
    o[so] = b2 + b5;
    o[2 * so] = b4 + b3;
    o[3 * so] = b6 + b1;
    o[4 * so] = b6 - b1;
    o[5 * so] = b4 - b3;
    o[6 * so] = b2 - b5;
    o[7 * so] = b0 - b7;
}

void x264o_add8x8_idct8(pixel *dst, int sd, const dctcoef d[64])
{
    int c[64], f[64], g[64];
    for (int i = 0; i < 64; i++) c[i] = d[i];
    for (int y = 0; y < 8; y++) inv8_1d(c + y * 8, 1, f + y * 8, 1);
    for (int x = 0; x < 8; x++) inv8_1d(f + x, 8, g + x, 8);
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            dst[y * sd + x] = clip_pixel(dst[y * sd + x] + ((g[y * 8 + x] + 32) >> 6));
}

/* 4-point Hadamard in H.264 sequency order: rows of [[1,1,1,1],[1,1,-1,-1],[1,-1,-1,1],[1,-1,1,-1]] */
static void had4_1d(const int *in, int si, int *out, int so)
{
    int s01 = in[0] + in[si], d01 = in[0] - in[si];
    int s23 = in[2 * si] + in[3 * si], d23 = in[2 * si] - in[3 * si];
    out[0] = s01 + s23;
    out[so] = s01 - s23;
    out[2 * so] = d01 - d23;
    out[3 * so] = d01 + d23;
}

static void had4x4(dctcoef d[16], int fwd)
{
    int c[16], t[16], o[16];
    for (int i = 0; i < 16; i++) c[i] = d[i];
    for (int y = 0; y < 4; y++) had4_1d(c + y * 4, 1, t + y * 4, 1);
    for (int x = 0; x < 4; x++) had4_1d(t + x, 4, o + x, 4);
    for (int i = 0; i < 16; i++) d[i] = (dctcoef)(fwd ? (o[i] + 1) >> 1 : o[i]);
}

void x264o_dct4x4dc(dctcoef d[16]) { had4x4(d, 1); }
void x264o_idct4x4dc(dctcoef d[16]) { had4x4(d, 0); }

void x264o_dct2x2dc(dctcoef d[4])
{
    int a = d[0] + d[1], b = d[0] - d[1], c = d[2] + d[3], e = d[2] - d[3];
    d[0] = (dctcoef)(a + c);
    d[1] = (dctcoef)(b + e);
    d[2] = (dctcoef)(a - c);
    d[3] = (dctcoef)(b - e);
}

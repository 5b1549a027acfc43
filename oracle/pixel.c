/* oracle/pixel.c — block distortion metrics (TEST INFRASTRUCTURE; see x264o.h header).
 * Restates [x264-upstream] common/pixel.c as described in SURVEY.md Appendix C; reached in the
 * reference only through x264_encoder_encode() (codec.c:1693).  parity unpinned vs libx264. */
#include "x264o.h"
#include <stdlib.h>

int x264o_sad(const pixel *a, int sa, const pixel *b, int sb, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            s += abs(a[y * sa + x] - b[y * sb + x]);
    return s;
}

int x264o_ssd(const pixel *a, int sa, const pixel *b, int sb, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int d = a[y * sa + x] - b[y * sb + x];
            s += d * d;
        }
    return s;
}

/* 1-D Hadamard butterflies on n (4 or 8) values with stride st */
static void hadamard_1d(int *v, int st, int n)
{
    for (int half = 1; half < n; half <<= 1)
        for (int i = 0; i < n; i += half * 2)
            for (int j = i; j < i + half; j++) {
                int p = v[j * st], q = v[(j + half) * st];
                v[j * st] = p + q;
                v[(j + half) * st] = p - q;
            }
}

static int hadamard_abs_sum(const pixel *a, int sa, const pixel *b, int sb, int n)
{
    int m[64];
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            m[y * n + x] = a[y * sa + x] - b[y * sb + x];
    for (int y = 0; y < n; y++) hadamard_1d(m + y * n, 1, n);
    for (int x = 0; x < n; x++) hadamard_1d(m + x, n, n);
    int s = 0;
    for (int i = 0; i < n * n; i++) s += abs(m[i]);
    return s;
}

/* SATD: per 4x4 sub-block, sum|H d H| >> 1 (always even, so the shift is exact per block) */
int x264o_satd(const pixel *a, int sa, const pixel *b, int sb, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4)
            s += hadamard_abs_sum(a + y * sa + x, sa, b + y * sb + x, sb, 4) >> 1;
    return s;
}

int x264o_sa8d_8x8_raw(const pixel *a, int sa, const pixel *b, int sb)
{
    return hadamard_abs_sum(a, sa, b, sb, 8);
}

/* SA8D: 8x8 -> (raw+2)>>2 ; 16x16 -> (sum of four raw + 2)>>2 */
int x264o_sa8d(const pixel *a, int sa, const pixel *b, int sb, int w, int h)
{
    int s = 0;
    for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8)
            s += x264o_sa8d_8x8_raw(a + y * sa + x, sa, b + y * sb + x, sb);
    return (s + 2) >> 2;
}

uint64_t x264o_var(const pixel *p, int stride, int w, int h)
{
    uint32_t sum = 0, sqr = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            sum += p[y * stride + x];
            sqr += p[y * stride + x] * p[y * stride + x];
        }
    return sum + ((uint64_t)sqr << 32);
}

/* hadamard_ac (psy-RD energy): AC energy of the 4x4 and 8x8 Hadamard transforms of the source
 * block itself (no prediction).  Returns ((sum8 >> 2) << 32) | (sum4 >> 1), DC terms removed. */
uint64_t x264o_hadamard_ac(const pixel *p, int stride, int w, int h)
{
    static const pixel zero[8] = {0};
    uint32_t sum4 = 0, sum8 = 0;
    for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8) {
            const pixel *q = p + y * stride + x;
            int dc4 = 0, dc8 = 0;
            for (int by = 0; by < 8; by += 4)
                for (int bx = 0; bx < 8; bx += 4) {
                    int m[16];
                    for (int j = 0; j < 4; j++)
                        for (int i = 0; i < 4; i++) m[j * 4 + i] = q[(by + j) * stride + bx + i];
                    for (int j = 0; j < 4; j++) hadamard_1d(m + j * 4, 1, 4);
                    for (int i = 0; i < 4; i++) hadamard_1d(m + i, 4, 4);
                    for (int i = 0; i < 16; i++) sum4 += abs(m[i]);
                    dc4 += abs(m[0]);
                }
            /* 8x8 transform of the raw pixels == hadamard_abs_sum against a zero block */
            {
                int m[64];
                for (int j = 0; j < 8; j++)
                    for (int i = 0; i < 8; i++) m[j * 8 + i] = q[j * stride + i] - zero[i];
                for (int j = 0; j < 8; j++) hadamard_1d(m + j * 8, 1, 8);
                for (int i = 0; i < 8; i++) hadamard_1d(m + i, 8, 8);
                for (int i = 0; i < 64; i++) sum8 += abs(m[i]);
                dc8 = abs(m[0]);
            }
            sum4 -= dc4;
            sum8 -= dc8;
        }
    return ((uint64_t)(sum8 >> 2) << 32) | (sum4 >> 1);
}

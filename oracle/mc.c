/* oracle/mc.c — interpolation & plane filters (TEST INFRASTRUCTURE; see x264o.h header).
 * Normative sample interpolation: ITU-T H.264 8.4.2.2.1 (luma 6-tap + quarter-sample averaging)
 * and 8.4.2.2.2 (chroma 1/8 bilinear).  Organisation as four half-pel planes + lowres planes
 * restates [x264-upstream] common/mc.c, common/frame.c, behind codec.c:1693. */
#include "x264o.h"

static inline pixel clip_pixel(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

static inline int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * b + 20 * c + 20 * d - 5 * e + f; }

/* half-pel samples right (H), below (V) and below-right (centre) of integer sample (x,y);
 * source reads are clamped to the w x h picture, which is exactly what edge replication gives */
void x264o_hpel_filter_pixel(const pixel *src, int stride, int w, int h, int x, int y,
                             pixel *ph, pixel *pv, pixel *pc)
{
#define S(xx, yy) src[clampi(yy, 0, h - 1) * stride + clampi(xx, 0, w - 1)]
    int hh = tap6(S(x - 2, y), S(x - 1, y), S(x, y), S(x + 1, y), S(x + 2, y), S(x + 3, y));
    int vv = tap6(S(x, y - 2), S(x, y - 1), S(x, y), S(x, y + 1), S(x, y + 2), S(x, y + 3));
    int col[6];
    for (int i = 0; i < 6; i++) {
        int xx = x - 2 + i;
        col[i] = tap6(S(xx, y - 2), S(xx, y - 1), S(xx, y), S(xx, y + 1), S(xx, y + 2), S(xx, y + 3));
    }
    int cc = tap6(col[0], col[1], col[2], col[3], col[4], col[5]);
    *ph = clip_pixel((hh + 16) >> 5);
    *pv = clip_pixel((vv + 16) >> 5);
    *pc = clip_pixel((cc + 512) >> 10);
#undef S
}

void x264o_frame_filter(pixel *plane[4], int stride, int w, int h, int pad)
{
    const pixel *src = plane[0];
    /* interpolated planes first (they read only the interior of plane[0]) */
    for (int y = -pad; y < h + pad; y++)
        for (int x = -pad; x < w + pad; x++)
            x264o_hpel_filter_pixel(src, stride, w, h, x, y, &plane[1][y * stride + x],
                                    &plane[2][y * stride + x], &plane[3][y * stride + x]);
    /* then replicate the full-pel border */
    for (int y = -pad; y < h + pad; y++)
        for (int x = -pad; x < w + pad; x++)
            if (x < 0 || x >= w || y < 0 || y >= h)
                plane[0][y * stride + x] = src[clampi(y, 0, h - 1) * stride + clampi(x, 0, w - 1)];
}

/* cascaded rounding averages (not a true 4-tap mean) — SURVEY.md Appendix C "Lowres planes" */
static inline int avg4(int a, int b, int c, int d) { return (((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1; }

void x264o_frame_init_lowres(const pixel *src, int ss, int w, int h, pixel *dst[4], int ds)
{
#define S(xx, yy) src[clampi(yy, 0, h - 1) * ss + clampi(xx, 0, w - 1)]
    for (int y = 0; y < h / 2; y++)
        for (int x = 0; x < w / 2; x++) {
            int X = 2 * x, Y = 2 * y;
            dst[0][y * ds + x] = (pixel)avg4(S(X, Y), S(X, Y + 1), S(X + 1, Y), S(X + 1, Y + 1));
            dst[1][y * ds + x] = (pixel)avg4(S(X + 1, Y), S(X + 1, Y + 1), S(X + 2, Y), S(X + 2, Y + 1));
            dst[2][y * ds + x] = (pixel)avg4(S(X, Y + 1), S(X, Y + 2), S(X + 1, Y + 1), S(X + 1, Y + 2));
            dst[3][y * ds + x] = (pixel)avg4(S(X + 1, Y + 1), S(X + 1, Y + 2), S(X + 2, Y + 1), S(X + 2, Y + 2));
        }
#undef S
}

void x264o_pixel_avg(pixel *dst, int sd, const pixel *a, int sa, const pixel *b, int sb, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * sd + x] = (pixel)((a[y * sa + x] + b[y * sb + x] + 1) >> 1);
}

/* bi-prediction of two motion-compensated blocks ([x264-upstream] common/mc.c pixel_avg_wxh / pixel_avg_weight_wxh, the B-frame and
 * weighted bi-pred path of A10): weight1 == 32 is the rounding average, otherwise (a * w1 + b * (64 - w1) + 32) >> 6, clipped */
void x264o_pixel_avg_weight(pixel *dst, int sd, const pixel *a, int sa, const pixel *b, int sb, int w, int h, int weight1)
{
    if (weight1 == 32) { x264o_pixel_avg(dst, sd, a, sa, b, sb, w, h); return; }
    int weight2 = 64 - weight1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = (a[y * sa + x] * weight1 + b[y * sb + x] * weight2 + 32) >> 6;
            dst[y * sd + x] = (pixel)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* explicit weighted prediction of one reference ([x264-upstream] common/mc.c mc_weight; 8.4.2.3 with one direction):
 * denom >= 1: ((src * scale + (1 << (denom - 1))) >> denom) + offset, else src * scale + offset; clipped */
void x264o_mc_weight(pixel *dst, int sd, const pixel *src, int ss, int w, int h, int scale, int denom, int offset)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int v = denom >= 1 ? ((src[y * ss + x] * scale + (1 << (denom - 1))) >> denom) + offset : src[y * ss + x] * scale + offset;
            dst[y * sd + x] = (pixel)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
}

/* which half-pel plane(s) realise each of the 16 quarter-sample positions (Figure 8-4 of the spec:
 * a,c,d,n,e,g,p,r,f,i,k,q are rounding averages of the two nearest integer/half samples) */
static const uint8_t qpel_plane0[16] = { 0, 1, 1, 1, 0, 1, 1, 1, 2, 3, 3, 3, 0, 1, 1, 1 };
static const uint8_t qpel_plane1[16] = { 0, 0, 1, 0, 2, 2, 3, 2, 2, 2, 3, 2, 2, 2, 3, 2 };

void x264o_mc_luma(pixel *dst, int sd, pixel *const plane[4], int stride, int x, int y,
                   int mvx, int mvy, int w, int h)
{
    int idx = ((mvy & 3) << 2) | (mvx & 3);
    int base = (y + (mvy >> 2)) * stride + x + (mvx >> 2);
    const pixel *s0 = plane[qpel_plane0[idx]] + base + ((mvy & 3) == 3) * stride;
    if (idx & 5) {
        const pixel *s1 = plane[qpel_plane1[idx]] + base + ((mvx & 3) == 3);
        x264o_pixel_avg(dst, sd, s0, stride, s1, stride, w, h);
    } else {
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) dst[j * sd + i] = s0[j * stride + i];
    }
}

/* (x,y) = chroma-sample position of the block; mv in quarter luma samples = eighth chroma samples */
void x264o_mc_chroma(pixel *dstu, pixel *dstv, int sd, const pixel *nv12, int stride,
                     int x, int y, int mvx, int mvy, int w, int h)
{
    int dx = mvx & 7, dy = mvy & 7;
    int cA = (8 - dx) * (8 - dy), cB = dx * (8 - dy), cC = (8 - dx) * dy, cD = dx * dy;
    const pixel *s = nv12 + (y + (mvy >> 3)) * stride + 2 * (x + (mvx >> 3));
    for (int j = 0; j < h; j++, s += stride)
        for (int i = 0; i < w; i++) {
            dstu[j * sd + i] = (pixel)((cA * s[2 * i] + cB * s[2 * i + 2] + cC * s[stride + 2 * i] + cD * s[stride + 2 * i + 2] + 32) >> 6);
            dstv[j * sd + i] = (pixel)((cA * s[2 * i + 1] + cB * s[2 * i + 3] + cC * s[stride + 2 * i + 1] + cD * s[stride + 2 * i + 3] + 32) >> 6);
        }
}

/* oracle/predict.c — intra predictors (TEST INFRASTRUCTURE; see x264o.h header).
 * Normative: ITU-T H.264 8.3.1.2 (4x4), 8.3.2.2 (8x8 incl. reference filtering), 8.3.3 (16x16),
 * 8.3.4 (chroma).  Plays the role of [x264-upstream] common/predict.c behind codec.c:1693. */
#include "x264o.h"

static inline pixel clip_pixel(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

static void fill(pixel *dst, int sd, int w, int h, int v)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) dst[y * sd + x] = (pixel)v;
}

/* generic V/H/plane for square blocks of size n (16 or 8-chroma) */
static void pred_v(pixel *dst, int sd, const pixel *src, int ss, int n)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[y * sd + x] = src[-ss + x];
}
static void pred_h(pixel *dst, int sd, const pixel *src, int ss, int n)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) dst[y * sd + x] = src[y * ss - 1];
}
static void pred_plane(pixel *dst, int sd, const pixel *src, int ss, int n)
{
    int half = n / 2, H = 0, V = 0;
    for (int i = 0; i < half; i++) {
        H += (i + 1) * (src[-ss + half + i] - src[-ss + half - 2 - i]);
        V += (i + 1) * (src[(half + i) * ss - 1] - src[(half - 2 - i) * ss - 1]);
    }
    int a = 16 * (src[(n - 1) * ss - 1] + src[-ss + n - 1]);
    int b = n == 16 ? (5 * H + 32) >> 6 : (34 * H + 32) >> 6;
    int c = n == 16 ? (5 * V + 32) >> 6 : (34 * V + 32) >> 6;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++)
            dst[y * sd + x] = clip_pixel((a + b * (x - (half - 1)) + c * (y - (half - 1)) + 16) >> 5);
}

void x264o_predict_16x16(pixel *dst, int sd, const pixel *src, int ss, int mode)
{
    int st = 0, sl = 0;
    switch (mode) {
    case I_PRED_16x16_V: pred_v(dst, sd, src, ss, 16); return;
    case I_PRED_16x16_H: pred_h(dst, sd, src, ss, 16); return;
    case I_PRED_16x16_P: pred_plane(dst, sd, src, ss, 16); return;
    case I_PRED_16x16_DC_128: fill(dst, sd, 16, 16, 128); return;
    default: break;
    }
    if (mode == I_PRED_16x16_DC || mode == I_PRED_16x16_DC_TOP)
        for (int i = 0; i < 16; i++) st += src[-ss + i];
    if (mode == I_PRED_16x16_DC || mode == I_PRED_16x16_DC_LEFT)
        for (int i = 0; i < 16; i++) sl += src[i * ss - 1];
    fill(dst, sd, 16, 16, mode == I_PRED_16x16_DC ? (st + sl + 16) >> 5 : (st + sl + 8) >> 4);
}

void x264o_predict_8x8c(pixel *dst, int sd, const pixel *src, int ss, int mode)
{
    switch (mode) {
    case I_PRED_CHROMA_V: pred_v(dst, sd, src, ss, 8); return;
    case I_PRED_CHROMA_H: pred_h(dst, sd, src, ss, 8); return;
    case I_PRED_CHROMA_P: pred_plane(dst, sd, src, ss, 8); return;
    case I_PRED_CHROMA_DC_128: fill(dst, sd, 8, 8, 128); return;
    default: break;
    }
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0; /* top-left4, top-right4, left-upper4, left-lower4 */
    if (mode != I_PRED_CHROMA_DC_LEFT)
        for (int i = 0; i < 4; i++) { s0 += src[-ss + i]; s1 += src[-ss + 4 + i]; }
    if (mode != I_PRED_CHROMA_DC_TOP)
        for (int i = 0; i < 4; i++) { s2 += src[i * ss - 1]; s3 += src[(4 + i) * ss - 1]; }
    int dc[4];
    if (mode == I_PRED_CHROMA_DC) {
        dc[0] = (s0 + s2 + 4) >> 3; dc[1] = (s1 + 2) >> 2; dc[2] = (s3 + 2) >> 2; dc[3] = (s1 + s3 + 4) >> 3;
    } else if (mode == I_PRED_CHROMA_DC_LEFT) {
        dc[0] = dc[1] = (s2 + 2) >> 2; dc[2] = dc[3] = (s3 + 2) >> 2;
    } else {
        dc[0] = dc[2] = (s0 + 2) >> 2; dc[1] = dc[3] = (s1 + 2) >> 2;
    }
    for (int b = 0; b < 4; b++) fill(dst + (b >> 1) * 4 * sd + (b & 1) * 4, sd, 4, 4, dc[b]);
}

/* e[] is the geometric edge line: e[C-1-y] = p[-1,y], e[C] = p[-1,-1], e[C+1+x] = p[x,-1] */
#define TOPN(x) e[C + 1 + (x)]
#define LEFTN(y) e[C - 1 - (y)]
#define F2(a, b) (((a) + (b) + 1) >> 1)
#define F3(a, b, c) (((a) + 2 * (b) + (c) + 2) >> 2)

/* directional modes shared by 4x4 and 8x8 (n = block size, e = edge line with corner at C) */
static void pred_dir(pixel *dst, int sd, const pixel *e, int C, int n, int mode)
{
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            int v;
            switch (mode) {
            case I_PRED_4x4_DDL:
                v = (x == n - 1 && y == n - 1) ? (TOPN(2 * n - 2) + 3 * TOPN(2 * n - 1) + 2) >> 2
                                               : F3(TOPN(x + y), TOPN(x + y + 1), TOPN(x + y + 2));
                break;
            case I_PRED_4x4_DDR:
                if (x > y) v = F3(TOPN(x - y - 2), TOPN(x - y - 1), TOPN(x - y));
                else if (x < y) v = F3(LEFTN(y - x - 2), LEFTN(y - x - 1), LEFTN(y - x));
                else v = F3(TOPN(0), TOPN(-1), LEFTN(0));
                break;
            case I_PRED_4x4_VR: {
                int z = 2 * x - y, k = x - (y >> 1);
                if (z >= 0 && !(z & 1)) v = F2(TOPN(k - 1), TOPN(k));
                else if (z > 0) v = F3(TOPN(k - 2), TOPN(k - 1), TOPN(k));
                else if (z == -1) v = F3(LEFTN(0), TOPN(-1), TOPN(0));
                else v = F3(LEFTN(y - 2 * x - 1), LEFTN(y - 2 * x - 2), LEFTN(y - 2 * x - 3));
                break; }
            case I_PRED_4x4_HD: {
                int z = 2 * y - x, k = y - (x >> 1);
                if (z >= 0 && !(z & 1)) v = F2(LEFTN(k - 1), LEFTN(k));
                else if (z > 0) v = F3(LEFTN(k - 2), LEFTN(k - 1), LEFTN(k));
                else if (z == -1) v = F3(LEFTN(0), TOPN(-1), TOPN(0));
                else v = F3(TOPN(x - 2 * y - 1), TOPN(x - 2 * y - 2), TOPN(x - 2 * y - 3));
                break; }
            case I_PRED_4x4_VL: {
                int k = x + (y >> 1);
                v = (y & 1) ? F3(TOPN(k), TOPN(k + 1), TOPN(k + 2)) : F2(TOPN(k), TOPN(k + 1));
                break; }
            default: { /* HU */
                int z = x + 2 * y, k = y + (x >> 1), last = 2 * n - 3; /* 5 for 4x4, 13 for 8x8 */
                if (z > last) v = LEFTN(n - 1);
                else if (z == last) v = (LEFTN(n - 2) + 3 * LEFTN(n - 1) + 2) >> 2;
                else if (z & 1) v = F3(LEFTN(k), LEFTN(k + 1), LEFTN(k + 2));
                else v = F2(LEFTN(k), LEFTN(k + 1));
                break; }
            }
            dst[y * sd + x] = (pixel)v;
        }
}

void x264o_predict_4x4(pixel *dst, int sd, const pixel *src, int ss, int mode, int avail)
{
    enum { C = 4 };
    pixel e[13];
    int st = 0, sl = 0;
    /* gather only what the mode may legally touch (unavailable samples are never read) */
    int need_top = mode == I_PRED_4x4_V || mode == I_PRED_4x4_DC || mode == I_PRED_4x4_DC_TOP ||
                   (mode >= I_PRED_4x4_DDL && mode <= I_PRED_4x4_VL);
    int need_left = mode == I_PRED_4x4_H || mode == I_PRED_4x4_DC || mode == I_PRED_4x4_DC_LEFT ||
                    mode == I_PRED_4x4_DDR || mode == I_PRED_4x4_VR || mode == I_PRED_4x4_HD ||
                    mode == I_PRED_4x4_HU;
    int need_tl = mode == I_PRED_4x4_DDR || mode == I_PRED_4x4_VR || mode == I_PRED_4x4_HD;
    for (int i = 0; i < 13; i++) e[i] = 128;
    if (need_top) {
        for (int i = 0; i < 4; i++) TOPN(i) = src[-ss + i];
        for (int i = 4; i < 8; i++) TOPN(i) = (avail & X264O_AVAIL_TOPRIGHT) ? src[-ss + i] : src[-ss + 3];
    }
    if (need_left) for (int i = 0; i < 4; i++) LEFTN(i) = src[i * ss - 1];
    if (need_tl) e[C] = src[-ss - 1];
    switch (mode) {
    case I_PRED_4x4_V: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * sd + x] = TOPN(x); return;
    case I_PRED_4x4_H: for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) dst[y * sd + x] = LEFTN(y); return;
    case I_PRED_4x4_DC_128: fill(dst, sd, 4, 4, 128); return;
    case I_PRED_4x4_DC: case I_PRED_4x4_DC_LEFT: case I_PRED_4x4_DC_TOP:
        if (mode != I_PRED_4x4_DC_LEFT) for (int i = 0; i < 4; i++) st += TOPN(i);
        if (mode != I_PRED_4x4_DC_TOP) for (int i = 0; i < 4; i++) sl += LEFTN(i);
        fill(dst, sd, 4, 4, mode == I_PRED_4x4_DC ? (st + sl + 4) >> 3 : (st + sl + 2) >> 2);
        return;
    default: pred_dir(dst, sd, e, C, 4, mode); return;
    }
}

/* 8.3.2.2.1 reference sample filtering.  edge layout: corner at index 15
 * (edge[14-y] = p'[-1,y], edge[15] = p'[-1,-1], edge[16+x] = p'[x,-1], x = 0..15). */
void x264o_predict_8x8_filter(const pixel *src, int ss, pixel edge[33], int avail)
{
    enum { C = 15 };
    pixel *e = edge;
    int top[16], left[8], tl = 128;
    int has_t = !!(avail & X264O_AVAIL_TOP), has_l = !!(avail & X264O_AVAIL_LEFT);
    int has_tl = !!(avail & X264O_AVAIL_TOPLEFT), has_tr = !!(avail & X264O_AVAIL_TOPRIGHT);
    for (int i = 0; i < 33; i++) edge[i] = 128;
    if (has_t) {
        for (int i = 0; i < 8; i++) top[i] = src[-ss + i];
        for (int i = 8; i < 16; i++) top[i] = has_tr ? src[-ss + i] : src[-ss + 7];
    }
    if (has_l) for (int i = 0; i < 8; i++) left[i] = src[i * ss - 1];
    if (has_tl) tl = src[-ss - 1];
    if (has_t) {
        TOPN(0) = (pixel)(has_tl ? F3(tl, top[0], top[1]) : (3 * top[0] + top[1] + 2) >> 2);
        for (int i = 1; i < 15; i++) TOPN(i) = (pixel)F3(top[i - 1], top[i], top[i + 1]);
        TOPN(15) = (pixel)((top[14] + 3 * top[15] + 2) >> 2);
    }
    if (has_l) {
        LEFTN(0) = (pixel)(has_tl ? F3(tl, left[0], left[1]) : (3 * left[0] + left[1] + 2) >> 2);
        for (int i = 1; i < 7; i++) LEFTN(i) = (pixel)F3(left[i - 1], left[i], left[i + 1]);
        LEFTN(7) = (pixel)((left[6] + 3 * left[7] + 2) >> 2);
    }
    if (has_tl) {
        if (has_t && has_l) e[C] = (pixel)F3(top[0], tl, left[0]);
        else if (has_t) e[C] = (pixel)((3 * tl + top[0] + 2) >> 2);
        else if (has_l) e[C] = (pixel)((3 * tl + left[0] + 2) >> 2);
        else e[C] = (pixel)tl;
    }
}

void x264o_predict_8x8(pixel *dst, int sd, const pixel edge[33], int mode)
{
    enum { C = 15 };
    const pixel *e = edge;
    int st = 0, sl = 0;
    switch (mode) {
    case I_PRED_4x4_V: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dst[y * sd + x] = TOPN(x); return;
    case I_PRED_4x4_H: for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) dst[y * sd + x] = LEFTN(y); return;
    case I_PRED_4x4_DC_128: fill(dst, sd, 8, 8, 128); return;
    case I_PRED_4x4_DC: case I_PRED_4x4_DC_LEFT: case I_PRED_4x4_DC_TOP:
        if (mode != I_PRED_4x4_DC_LEFT) for (int i = 0; i < 8; i++) st += TOPN(i);
        if (mode != I_PRED_4x4_DC_TOP) for (int i = 0; i < 8; i++) sl += LEFTN(i);
        fill(dst, sd, 8, 8, mode == I_PRED_4x4_DC ? (st + sl + 8) >> 4 : (st + sl + 4) >> 3);
        return;
    default: pred_dir(dst, sd, e, C, 8, mode); return;
    }
}

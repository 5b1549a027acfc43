/* oracle/deblock.c — in-loop deblocking edge filters (TEST INFRASTRUCTURE; see x264o.h header).
 * Normative: ITU-T H.264 8.7.2.2-8.7.2.4 (tables 8-16/8-17).  Plays the role of
 * [x264-upstream] common/deblock.c behind x264_encoder_encode() (codec.c:1693). */
#include "x264o.h"
#include <stdlib.h>

const uint8_t x264o_alpha_table[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
    15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
const uint8_t x264o_beta_table[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
const uint8_t x264o_tc0_table[52][3] = {
    { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
    { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
    { 0, 0, 0 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 }, { 0, 1, 1 }, { 1, 1, 1 },
    { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 2, 3 },
    { 1, 2, 3 }, { 2, 2, 3 }, { 2, 2, 4 }, { 2, 3, 4 }, { 2, 3, 4 }, { 3, 3, 5 }, { 3, 4, 6 }, { 3, 4, 6 },
    { 4, 5, 7 }, { 4, 5, 8 }, { 4, 6, 9 }, { 5, 7, 10 }, { 6, 8, 11 }, { 6, 8, 13 }, { 7, 10, 14 }, { 8, 11, 16 },
    { 9, 12, 18 }, { 10, 13, 20 }, { 11, 15, 23 }, { 13, 17, 25 } };

static inline pixel clip_pixel(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }
static inline int clip3(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* pix points at q0; p-side samples are at pix[-1*xstride], pix[-2*xstride], ... */
void x264o_deblock_luma_edge(pixel *pix, int xs, int ys, int lines, int alpha, int beta, int tc0, int bs)
{
    if (!bs) return;
    for (int l = 0; l < lines; l++, pix += ys) {
        int p2 = pix[-3 * xs], p1 = pix[-2 * xs], p0 = pix[-xs];
        int q0 = pix[0], q1 = pix[xs], q2 = pix[2 * xs];
        if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) continue;
        int ap = abs(p2 - p0), aq = abs(q2 - q0);
        if (bs < 4) {
            int tc = tc0 + (ap < beta) + (aq < beta);
            int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            if (ap < beta) pix[-2 * xs] = (pixel)(p1 + clip3((p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1, -tc0, tc0));
            if (aq < beta) pix[xs] = (pixel)(q1 + clip3((q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1, -tc0, tc0));
            pix[-xs] = clip_pixel(p0 + delta);
            pix[0] = clip_pixel(q0 - delta);
        } else {
            int strong = abs(p0 - q0) < ((alpha >> 2) + 2);
            if (ap < beta && strong) {
                int p3 = pix[-4 * xs];
                pix[-xs] = (pixel)((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3);
                pix[-2 * xs] = (pixel)((p2 + p1 + p0 + q0 + 2) >> 2);
                pix[-3 * xs] = (pixel)((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3);
            } else
                pix[-xs] = (pixel)((2 * p1 + p0 + q1 + 2) >> 2);
            if (aq < beta && strong) {
                int q3 = pix[3 * xs];
                pix[0] = (pixel)((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3);
                pix[xs] = (pixel)((p0 + q0 + q1 + q2 + 2) >> 2);
                pix[2 * xs] = (pixel)((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3);
            } else
                pix[0] = (pixel)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}

void x264o_deblock_chroma_edge(pixel *pix, int xs, int ys, int lines, int alpha, int beta, int tc0, int bs)
{
    if (!bs) return;
    for (int l = 0; l < lines; l++, pix += ys) {
        int p1 = pix[-2 * xs], p0 = pix[-xs], q0 = pix[0], q1 = pix[xs];
        if (abs(p0 - q0) >= alpha || abs(p1 - p0) >= beta || abs(q1 - q0) >= beta) continue;
        if (bs < 4) {
            int tc = tc0 + 1;
            int delta = clip3((((q0 - p0) * 4) + (p1 - q1) + 4) >> 3, -tc, tc);
            pix[-xs] = clip_pixel(p0 + delta);
            pix[0] = clip_pixel(q0 - delta);
        } else {
            pix[-xs] = (pixel)((2 * p1 + p0 + q1 + 2) >> 2);
            pix[0] = (pixel)((2 * q1 + q0 + p1 + 2) >> 2);
        }
    }
}

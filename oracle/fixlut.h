/* oracle/fixlut.h — the tables and single-float helpers of the AQ / macroblock-tree arithmetic, computed HERE with libm (TEST INFRASTRUCTURE ONLY).
 * The product carries the same tables as literals (include/x264gpu_log2f_lut.inc, include/x264gpu_exp2_lut.inc); the oracle must not read
 * those files — a wrong literal would then go unnoticed.  tests/test_cpu_oracle.py compares the two derivations entry by entry.
 *   log2 table: x264_log2_lut — log2(1 + i / 128) as five-decimal literals, i = 0..127
 *   exp2 table: round(256 * (2^(k / 64) - 1)), k = 0..63 (x264_exp2_lut) */
#ifndef X264O_FIXLUT_H
#define X264O_FIXLUT_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
static inline const uint16_t *x264o_exp2_lut(void)
{
    static uint16_t t[64]; static int done;
    if (!done) { for (int k = 0; k < 64; k++) t[k] = (uint16_t)lround(256.0 * (pow(2.0, k / 64.0) - 1.0)); done = 1; }
    return t;
}
/* ---- x264's own SINGLE-FLOAT arithmetic of AQ / macroblock-tree / x264_ratecontrol_mb_qp ([x264-upstream] common/tables.c x264_log2_lut,
 * common/common.h x264_log2 / x264_exp2fix8), evaluated as the C source reads (no contraction: -ffp-contract=off).
 *   x264_log2_lut[i]  = the five-decimal literal of log2(1 + i / 128) as C reads it: a double literal stored in a float
 *   x264_log2(x)      = x264_log2_lut[(x << clz(x) >> 24) & 0x7f] + (31 - clz(x))
 *   x264_exp2fix8(x)  = i = (int)(x * (-64.f / 6.f) + 512.5f); 0 below 0, 0xffff above 1023, else (x264_exp2_lut[i & 63] + 256) << (i >> 6) >> 8 */
static inline const float *x264o_log2f_lut(void)
{
    static float t[128]; static int done;
    if (!done) { for (int i = 0; i < 128; i++) { char b[32]; snprintf(b, sizeof b, "%.5f", log2(1.0 + i / 128.0)); t[i] = (float)strtod(b, NULL); } done = 1; }
    return t;
}
static inline float x264o_log2(uint32_t x)
{
    const int lz = __builtin_clz(x);
    return x264o_log2f_lut()[((x << lz) >> 24) & 0x7f] + (float)(31 - lz);
}
static inline int x264o_exp2fix8(float x)
{
    const int i = (int)(x * (-64.f / 6.f) + 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (int)(((uint32_t)(x264o_exp2_lut()[i & 63] + 256) << (i >> 6)) >> 8);
}
/* x264_ratecontrol_mb_qp: clip3((int)(qpm + offset + 0.5f)): two float additions, truncation; the clip is the caller's */
static inline int x264o_mb_qp(float qpm, float offset) { float qp = qpm; qp += offset; return (int)(qp + 0.5f); }
#endif

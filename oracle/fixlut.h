/* oracle/fixlut.h — the two fixed-point tables of the AQ / macroblock-tree arithmetic, computed HERE with libm (TEST INFRASTRUCTURE ONLY).
 * The product carries the same tables as literals (include/x264gpu_aq_lut.inc, include/x264gpu_exp2_lut.inc); the oracle must not read
 * those files — a wrong literal would then go unnoticed.  tests/test_cpu_oracle.py compares the two derivations entry by entry.
 *   log2 table: round(256 * log2(1 + i / 128)), i = 0..127 (fraction of the fixed-point log2; x264's x264_log2_lut is the float of the same)
 *   exp2 table: round(256 * (2^(k / 64) - 1)), k = 0..63 (x264_exp2_lut) */
#ifndef X264O_FIXLUT_H
#define X264O_FIXLUT_H
#include <math.h>
#include <stdint.h>
static inline const uint8_t *x264o_log2_lut(void)
{
    static uint8_t t[128]; static int done;
    if (!done) { for (int i = 0; i < 128; i++) t[i] = (uint8_t)lround(256.0 * log2(1.0 + i / 128.0)); done = 1; }
    return t;
}
static inline const uint16_t *x264o_exp2_lut(void)
{
    static uint16_t t[64]; static int done;
    if (!done) { for (int k = 0; k < 64; k++) t[k] = (uint16_t)lround(256.0 * (pow(2.0, k / 64.0) - 1.0)); done = 1; }
    return t;
}
#endif

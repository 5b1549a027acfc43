// rd.hip.h — bit counts for the rate-distortion costs of the macroblock loop (k_mb.hip.h, RD instantiations): x264_macroblock_size_cavlc
// ([x264-upstream] encoder/cavlc.c compiled with RDO_SKIP_BS) counts exactly the bits the macroblock layer takes in a CAVLC slice.
// The code tables are the host writer's (host/cavlc_tables.hpp), placed in constant memory for the device.
#pragma once
#include "enc_common.hip.h"
#define CAVLC_TABLE static __constant__ const
#define CAVLC_NAMESPACE x264gpu_cavlc
#include "../host/cavlc_tables.hpp"
#undef CAVLC_TABLE
#undef CAVLC_NAMESPACE

namespace x264gpu {

__device__ __forceinline__ int bs_size_ue_d(int v) { return 2 * (31 - __builtin_clz((unsigned)(v + 1))) + 1; }
__device__ __forceinline__ int bs_size_se_d(int v) { return bs_size_ue_d(v <= 0 ? -2 * v : 2 * v - 1); }

__device__ __forceinline__ int cavlc_level_bits(int code, int suffix_len)
{
    if (suffix_len == 0) {
        if (code < 14) return code + 1;
        if (code < 30) return 19;
        code -= 30;
    } else {
        if ((code >> suffix_len) < 15) return (code >> suffix_len) + 1 + suffix_len;
        code -= 15 << suffix_len;
    }
    if (code < 4096) return 28;
    code -= 4096;
    int prefix = 16;
    while (code >= (1 << (prefix - 3))) { code -= 1 << (prefix - 3); prefix++; }
    return 2 * prefix - 2;
}

// bits of residual_block_cavlc (9.2) for the n coefficients l[0..n-1] (scan order, LDS or global), nC as derived by the caller (-1: chroma DC).
// One lane per block: lanes run their own loops.
__device__ __forceinline__ int cavlc_block_bits(const int16_t *l, int n, int nC)
{
    namespace T = x264gpu_cavlc;
    int lev[16];
    int total = 0, last = -1;
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { lev[i] = i < n ? (int)l[i] : 0; if (lev[i]) { mask |= 1u << i; total++; last = i; } }
    // trailing ones: up to three +-1 at the high-frequency end
    int t1 = 0;
    {
        unsigned m = mask;
        while (m && t1 < 3) { const int i = 31 - __builtin_clz(m); int v = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) v = q == i ? lev[q] : v;
            if (abs(v) != 1) break; t1++; m &= ~(1u << i); }
    }
    int bits = nC < 0 ? T::chroma_dc_coeff_token_len[4 * total + t1] : T::coeff_token_len[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3][4 * total + t1];
    if (!total) return bits;
    bits += t1;
    int suffix_len = total > 10 && t1 < 3 ? 1 : 0;
    {
        unsigned m = mask;
        for (int k = 0; k < t1; k++) m &= ~(1u << (31 - __builtin_clz(m)));          // drop the trailing ones
        bool first = true;
        while (m) {
            const int i = 31 - __builtin_clz(m);
            m &= ~(1u << i);
            int level = 0;
#pragma unroll
            for (int q = 0; q < 16; q++) level = q == i ? lev[q] : level;
            int code = level > 0 ? 2 * level - 2 : -2 * level - 1;
            if (first && t1 < 3) code -= 2;
            first = false;
            bits += cavlc_level_bits(code, suffix_len);
            if (suffix_len == 0) suffix_len = 1;
            if (abs(level) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
    }
    if (total < n) {
        const int zeros = last + 1 - total;
        bits += nC < 0 ? T::chroma_dc_total_zeros_len[total - 1][zeros] : T::total_zeros_len[total - 1][zeros];
        int left = zeros;
        unsigned m = mask;
        while (left > 0) {
            const int hi = 31 - __builtin_clz(m);
            m &= ~(1u << hi);
            if (!m) break;
            const int run = hi - (31 - __builtin_clz(m)) - 1;
            bits += T::run_before_len[(left < 7 ? left : 7) - 1][run];
            left -= run;
        }
    }
    return bits;
}

// bits of coded_block_pattern me(v) (Table 9-4)
__device__ __forceinline__ int cavlc_cbp_bits(int cbp, bool inter)
{
    return bs_size_ue_d(inter ? x264gpu_cavlc::cbp_to_golomb_inter[cbp] : x264gpu_cavlc::cbp_to_golomb_intra[cbp]);
}

}  // namespace x264gpu

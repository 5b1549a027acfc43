// oracle/rd.cpp — bit counts of CAVLC syntax for the rate-distortion costs of oracle/analyse.c (TEST INFRASTRUCTURE ONLY).
// x264_macroblock_size_cavlc ([x264-upstream] encoder/cavlc.c compiled with RDO_SKIP_BS) counts exactly the bits the writer would emit;
// this file holds the table-driven parts: residual_block_cavlc (9.2) and the coded_block_pattern code number (Table 9-4), on the checker
// decoder's own tables (cavlc_dec.hpp).
#include "cavlc_dec.hpp"
#include <cstdlib>

static int level_bits(int code, int suffix_len)
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
    return prefix + 1 + prefix - 3;
}

// bits of residual_block_cavlc for l[0..n-1] (scan order), nC as derived by the caller (-1: chroma DC); *total_out = total_coeff
extern "C" int x264o_cavlc_block_bits(const int16_t *l, int n, int nC, int *total_out)
{
    const cavlcdec::Tables &T = cavlcdec::tables();
    int idx[16], total = 0, bits;
    for (int i = 0; i < n; i++) if (l[i]) idx[total++] = i;
    int t1 = 0;
    for (int k = total - 1; k >= 0 && t1 < 3; k--) { if (abs(l[idx[k]]) == 1) t1++; else break; }
    if (nC < 0) bits = T.chroma_dc_coeff_token_len[4 * total + t1];
    else bits = T.coeff_token_len[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3][4 * total + t1];
    if (total_out) *total_out = total;
    if (!total) return bits;
    bits += t1;
    int suffix_len = total > 10 && t1 < 3 ? 1 : 0;
    for (int k = total - 1 - t1; k >= 0; k--) {
        const int level = l[idx[k]];
        int code = level > 0 ? 2 * level - 2 : -2 * level - 1;
        if (k == total - 1 - t1 && t1 < 3) code -= 2;
        bits += level_bits(code, suffix_len);
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    if (total < n) {
        const int zeros = idx[total - 1] + 1 - total;
        bits += nC < 0 ? T.chroma_dc_total_zeros_len[total - 1][zeros] : T.total_zeros_len[total - 1][zeros];
        int left = zeros;
        for (int k = total - 1; k > 0 && left > 0; k--) {
            const int run = idx[k] - idx[k - 1] - 1;
            bits += T.run_before_len[(left < 7 ? left : 7) - 1][run];
            left -= run;
        }
    }
    return bits;
}

// bits of coded_block_pattern me(v): ue of the code number of Table 9-4
extern "C" int x264o_cavlc_cbp_bits(int cbp, int inter)
{
    for (int code = 0; code < 48; code++)
        if (cavlcdec::kCbpOfCode[code][inter ? 1 : 0] == cbp) { int v = code + 1, n = 0; while (v >> n) n++; return 2 * n - 1; }
    return 99;
}

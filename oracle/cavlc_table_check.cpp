// oracle/cavlc_table_check.cpp — compares the two separately typed copies of the CAVLC code tables (TEST INFRASTRUCTURE ONLY): the
// product's (x264vfw_amd/host/cavlc_tables.hpp, (length, value) arrays) against the checker decoder's (cavlc_dec.hpp, bit strings), and
// checks that each of the decoder's tables is a prefix code.  Included here for comparison only; the decoder never reads the product's copy.
#include "cavlc_dec.hpp"
#include "../x264vfw_amd/host/cavlc_tables.hpp"

static int prefix_clash(const uint8_t *len, const uint16_t *bits, int n)
{
    int bad = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
            if (i != j && len[i] && len[j] && len[i] <= len[j] && (bits[j] >> (len[j] - len[i])) == bits[i]) bad++;
    return bad;
}

extern "C" int x264o_cavlc_tables_mismatches(void)
{
    const cavlcdec::Tables &T = cavlcdec::tables();
    int bad = 0;
    for (int c = 0; c < 4; c++) for (int i = 0; i < 68; i++) bad += T.coeff_token_len[c][i] != x264host::coeff_token_len[c][i] || T.coeff_token_bits[c][i] != x264host::coeff_token_bits[c][i];
    for (int i = 0; i < 20; i++) bad += T.chroma_dc_coeff_token_len[i] != x264host::chroma_dc_coeff_token_len[i] || T.chroma_dc_coeff_token_bits[i] != x264host::chroma_dc_coeff_token_bits[i];
    for (int t = 0; t < 15; t++) for (int i = 0; i < 16; i++) bad += T.total_zeros_len[t][i] != x264host::total_zeros_len[t][i] || T.total_zeros_bits[t][i] != x264host::total_zeros_bits[t][i];
    for (int t = 0; t < 3; t++) for (int i = 0; i < 4; i++) bad += T.chroma_dc_total_zeros_len[t][i] != x264host::chroma_dc_total_zeros_len[t][i] || T.chroma_dc_total_zeros_bits[t][i] != x264host::chroma_dc_total_zeros_bits[t][i];
    for (int t = 0; t < 7; t++) for (int i = 0; i < 16; i++) bad += T.run_before_len[t][i] != x264host::run_before_len[t][i] || T.run_before_bits[t][i] != x264host::run_before_bits[t][i];
    for (int code = 0; code < 48; code++) bad += x264host::cbp_to_golomb_intra[cavlcdec::kCbpOfCode[code][0]] != code || x264host::cbp_to_golomb_inter[cavlcdec::kCbpOfCode[code][1]] != code;
    return bad;
}

extern "C" int x264o_cavlc_tables_prefix_clashes(void)
{
    const cavlcdec::Tables &T = cavlcdec::tables();
    int bad = 0;
    for (int c = 0; c < 4; c++) bad += prefix_clash(T.coeff_token_len[c], T.coeff_token_bits[c], 68);
    bad += prefix_clash(T.chroma_dc_coeff_token_len, T.chroma_dc_coeff_token_bits, 20);
    for (int t = 0; t < 15; t++) bad += prefix_clash(T.total_zeros_len[t], T.total_zeros_bits[t], 16);
    for (int t = 0; t < 3; t++) bad += prefix_clash(T.chroma_dc_total_zeros_len[t], T.chroma_dc_total_zeros_bits[t], 4);
    for (int t = 0; t < 7; t++) bad += prefix_clash(T.run_before_len[t], T.run_before_bits[t], 16);
    return bad;
}

// oracle/cabac_table_check.cpp — compares the two separately typed copies of the CABAC context initialisation values (TEST INFRASTRUCTURE ONLY):
// the product's (x264vfw_amd/host/cabac_tables.hpp, per context index) against the checker decoder's (cabac_dec.hpp, per syntax element).
// Included here for comparison only; the decoder itself never reads the product's copy.
#include "cabac_dec.hpp"
#include "../x264vfw_amd/host/cabac_tables.hpp"

extern "C" int x264o_cabac_tables_mismatches(void)
{
    using namespace cabacdec;
    int bad = 0;
    auto row = [](int ctx) { return ctx < 276 ? x264host::cabac_init_0_275[ctx] : x264host::cabac_init_399_435[ctx - 399]; };
    auto chk = [&](int ctx, MN v, int pslice) {
        const x264host::CabacInitRow r = row(ctx);
        if (pslice ? (r.mp != v.m || r.np != v.n) : (r.mi != v.m || r.ni != v.n)) bad++;
    };
    for (int t = 0; t < 2; t++) {
        for (int i = 0; i < 8; i++) chk(3 + i, kMbTypeI[i], t);
        for (int i = 0; i < 4; i++) { chk(60 + i, kQpDelta[i], t); chk(64 + i, kChromaPred[i], t); chk(73 + i, kCbpLuma[t][i], t); }
        chk(68, kIntra4x4[0], t); chk(69, kIntra4x4[1], t);
        for (int i = 0; i < 8; i++) chk(77 + i, kCbpChroma[t][i], t);
        for (int i = 0; i < 20; i++) chk(85 + i, kCbf[t][i], t);
        for (int i = 0; i < 61; i++) { chk(105 + i, kSig[t][i], t); chk(166 + i, kLast[t][i], t); }
        for (int i = 0; i < 49; i++) chk(227 + i, kAbs[t][i], t);
        for (int i = 0; i < 3; i++) chk(399 + i, kT8[t][i], t);
        for (int i = 0; i < 15; i++) chk(402 + i, kSig8[t][i], t);
        for (int i = 0; i < 9; i++) chk(417 + i, kLast8[t][i], t);
        for (int i = 0; i < 10; i++) chk(426 + i, kAbs8[t][i], t);
    }
    for (int i = 0; i < 3; i++) { chk(11 + i, kSkipP[i], 1); chk(21 + i, kSubMbTypeP[i], 1); chk(24 + i, kSkipB[i], 1); }
    for (int i = 0; i < 9; i++) chk(27 + i, kMbTypeB[i], 1);
    for (int i = 0; i < 4; i++) chk(36 + i, kSubMbTypeB[i], 1);
    for (int i = 0; i < 7; i++) { chk(14 + i, kMbTypeP[i], 1); chk(40 + i, kMvdX[i], 1); chk(47 + i, kMvdY[i], 1); }
    for (int i = 0; i < 6; i++) chk(54 + i, kRefIdx[i], 1);
    for (int i = 0; i < 63; i++) if (kSigInc8[i] != x264host::cabac_sig8x8[i] || kLastInc8[i] != x264host::cabac_last8x8[i]) bad++;
    for (int i = 0; i < 64; i++) { if (kNextLps[i] != x264host::cabac_trans_lps[i]) bad++; for (int q = 0; q < 4; q++) if (kRangeLps[i][q] != x264host::cabac_range_lps[i][q]) bad++; }
    return bad;
}

// oracle/trellis.cpp — trellis quantisation as x264 runs it in CABAC sessions (TEST INFRASTRUCTURE ONLY; see x264o.h).
//
// [x264-upstream] encoder/rdo.c quant_trellis_cabac + trellis_coef* + trellis_dc_shortcut, reached from x264_macroblock_encode behind
// x264_encoder_encode (reference call site codec.c:1693); `--trellis 1` = the final encode of a macroblock only (preset medium).
// A Viterbi search over the levels of a block: the coefficients are visited from the last non-zero one of a round-to-nearest guess down,
// each may keep the guess q or take q - 1, a node is the abs-level context state CABAC would be in (eight of them), a path's score is
// weighted squared error + lambda2 x the bits the size-only coder (cabac_rd.cpp's tables) charges on the slice's live context variables.
// The statement of the algorithm is in oracle/TRELLIS_NOTES.md.  Restated from memory like the rest of the oracle: parity unpinned.
#include "cabac_dec.hpp"
#include <cmath>
#include <cstdlib>
#include <cstring>
extern "C" {
#include "encoder_priv.h"
}

namespace {

struct Tab {
    uint8_t trans[128][2];            // context variable after coding bin b: (pStateIdx << 1) | valMPS
    uint16_t size_unary[15][128];     // bits (1/256) of: prefix - 1 ones, a zero unless prefix == 14, the sign
    uint8_t trans_unary[15][128];
    Tab()
    {
        for (int st = 0; st < 128; st++) {
            const int s = st >> 1, mps = st & 1;
            trans[st][mps] = (uint8_t)(((s < 62 ? s + 1 : 62) << 1) | mps);
            trans[st][!mps] = (uint8_t)((cabacdec::kNextLps[s] << 1) | (s == 0 ? !mps : mps));
        }
        const uint16_t *ent = x264o_cabac_entropy();
        for (int prefix = 0; prefix < 15; prefix++)
            for (int ctx0 = 0; ctx0 < 128; ctx0++) {
                int bits = 0, ctx = ctx0;
                for (int i = 1; i < prefix; i++) { bits += ent[ctx ^ 1]; ctx = trans[ctx][1]; }
                if (prefix > 0 && prefix < 14) { bits += ent[ctx ^ 0]; ctx = trans[ctx][0]; }
                bits += 256;
                size_unary[prefix][ctx0] = (uint16_t)bits; trans_unary[prefix][ctx0] = (uint8_t)ctx;
            }
    }
};
const Tab &tab() { static const Tab t; return t; }
// cost of bin b on context variable st: x264_cabac_entropy[state ^ b] (the table is laid out [pStateIdx * 2 + (b != valMPS)])
inline int ent(int st, int b) { return x264o_cabac_entropy()[st ^ b]; }
inline int size_ue_big(unsigned v) { int n = 0; v++; while (v >> (n + 1)) n++; return 2 * n + 1; }
inline int sign_of(int v, int s) { return s < 0 ? -v : v; }            // SIGN(x, y)

const uint64_t SCORE_MAX = ~0ull;            // "negative": the node is dead
const uint64_t SCORE_BIAS = 1ull << 60;
inline bool live(uint64_t s) { return (int64_t)s >= 0; }

struct Node { uint64_t score; int level_idx; uint8_t cabac_state[4]; };     // the context variables a path can touch twice: 0, 4, 8, 9
struct Level { uint16_t next, abs_level; };

struct Trellis {
    Node nodes[2][8];
    Node *cur, *prev;
    Level tree[64 * 8 * 2 + 1];
    int used;
    uint8_t level_state[16];
    int lambda2;
    int cost_siglast[3];

    void set_level(Node &dst, const Node &src, int l) { tree[used].next = (uint16_t)src.level_idx; tree[used].abs_level = (uint16_t)l; dst.level_idx = used++; }

    // one value of one coefficient from source node j into node_ctx
    void coef(int j, int const_level, int abs_level, int prefix, int suffix_cost, int node_ctx, int level1_ctx, int levelgt1_ctx, uint64_t ssd)
    {
        uint64_t score = prev[j].score + ssd;
        unsigned f8 = (unsigned)cost_siglast[j ? 1 : 2];
        const uint8_t level1_state = j >= 3 ? prev[j].cabac_state[level1_ctx >> 2] : level_state[level1_ctx];
        f8 += ent(level1_state, const_level > 1);
        uint8_t levelgt1_state = 0;
        if (const_level > 1) {
            levelgt1_state = j >= 6 ? prev[j].cabac_state[levelgt1_ctx - 6] : level_state[levelgt1_ctx];
            f8 += tab().size_unary[prefix][levelgt1_state] + suffix_cost;
        } else f8 += 256;
        score += (uint64_t)f8 * lambda2 >> 4;
        if (score < cur[node_ctx].score) {              // strict: the earlier candidate keeps a tie
            cur[node_ctx].score = score;
            if (j == 2 || (j <= 3 && node_ctx == 4)) memcpy(cur[node_ctx].cabac_state, level_state + 12, 4);
            else if (j >= 3) memcpy(cur[node_ctx].cabac_state, prev[j].cabac_state, 4);
            if (j >= 3) cur[node_ctx].cabac_state[level1_ctx >> 2] = tab().trans[level1_state][const_level > 1];
            if (const_level > 1 && node_ctx == 7) cur[node_ctx].cabac_state[levelgt1_ctx - 6] = tab().trans_unary[prefix][levelgt1_state];
            cur[node_ctx].level_idx = prev[j].level_idx;
            set_level(cur[node_ctx], prev[j], abs_level);
        }
    }
    void coef0(int ctx_hi, uint64_t ssd0)
    {
        if (!ctx_hi) {
            cur[0].score = prev[0].score + ssd0; cur[0].level_idx = prev[0].level_idx;
            for (int j = 1; j < 4 && live(prev[j].score); j++) {
                cur[j].score = prev[j].score;
                if (j >= 3) memcpy(cur[j].cabac_state, prev[j].cabac_state, 4);
                set_level(cur[j], prev[j], 0);
            }
        } else
            for (int j = 1; j < 8; j++)
                if (live(prev[j].score)) {
                    cur[j].score = prev[j].score;
                    if (j >= 3) memcpy(cur[j].cabac_state, prev[j].cabac_state, 4);
                    set_level(cur[j], prev[j], 0);
                }
    }
    // level 1 / level >= 2 from every live source node, ascending; in ctx_lo the live nodes are contiguous from 0
    void coef1(int ctx_hi, uint64_t ssd0, uint64_t ssd1)
    {
        static const int8_t to[8] = { 1, 2, 3, 3, 4, 5, 6, 7 }, l1[8] = { 1, 2, 3, 4, 0, 0, 0, 0 };
        for (int j = ctx_hi; j < (ctx_hi ? 8 : 4); j++) {
            if (!j || live(prev[j].score)) coef(j, 1, 1, 1, 0, to[j], l1[j], 0, j ? ssd1 : ssd0);
            else if (!ctx_hi) return;
        }
    }
    void coefn(int ctx_hi, int abs_level, uint64_t ssd0, uint64_t ssd1, int levelgt1_last)
    {
        static const int8_t to[8] = { 4, 4, 4, 4, 5, 6, 7, 7 }, l1[8] = { 1, 2, 3, 4, 0, 0, 0, 0 }, lg[8] = { 5, 5, 5, 5, 6, 7, 8, 9 };
        const int prefix = abs_level - 1 < 14 ? abs_level - 1 : 14;
        const int suffix_cost = abs_level >= 15 ? size_ue_big((unsigned)(abs_level - 15)) << 8 : 0;
        for (int j = ctx_hi; j < (ctx_hi ? 8 : 4); j++) {
            if (!j || live(prev[j].score)) coef(j, 2, abs_level, prefix, suffix_cost, to[j], l1[j], j == 7 ? levelgt1_last : lg[j], j ? ssd1 : ssd0);
            else if (!ctx_hi) return;
        }
    }
};

const uint32_t kW4[3] = { 800, 320, 128 };                                  // FIX8(3.125), FIX8(1.25), FIX8(0.5)
const uint32_t kW8[6] = { 256, 201, 656, 227, 410, 363 };                   // FIX8 of 1.00000, 0.78487, 2.56132, 0.88637, 1.60040, 1.41850
inline uint32_t weight4(int pos) { return kW4[(pos & 1) + ((pos >> 2) & 1)]; }
inline uint32_t weight8(int pos)
{
    static const uint8_t cls[4][4] = { { 0, 3, 4, 3 }, { 3, 1, 5, 1 }, { 4, 5, 2, 5 }, { 3, 1, 5, 1 } };
    return kW8[cls[(pos >> 3) & 3][pos & 3]];
}
int trellis_lambda2(int intra, int qp)
{
    return (int)((intra ? 0.65 * 0.65 : 0.85 * 0.85) * pow(2.0, qp / 3.0 + 6.0) + 0.5);        // x264_trellis_lambda2_tab: 46, 58, 73, ... / 27, 34, 43, ...
}

}  // namespace

extern "C" int x264o_unquant4(int qp, int pos);
extern "C" int x264o_unquant8(int qp, int pos);

// dct: the block's transform coefficients in raster order (chroma DC: the 2x2 in its own order); replaced by the levels.  cat: the CABAC
// block category (0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8).  mf: the quantiser row of the block's list at
// this qp.  state: the slice's 460 context variables (read only).  Returns whether a non-zero level is left.
extern "C" int x264o_quant_trellis_cabac(dctcoef *dct, const uint16_t *mf, int qp, int cat, int intra, const uint8_t *state)
{
    static const int sig_off[6] = { 105, 120, 134, 149, 152, 402 }, last_off[6] = { 166, 181, 195, 210, 213, 417 }, abs_off[6] = { 227, 237, 247, 257, 266, 426 };
    const int num_coefs = cat == 5 ? 64 : cat == 3 ? 4 : 16, b_ac = cat == 1 || cat == 4, dc = cat == 0 || cat == 3, b_chroma = cat == 3 || cat == 4;
    const int lambda2 = trellis_lambda2(intra, qp);
    const uint8_t *state_sig = state + sig_off[cat], *state_last = state + last_off[cat], *cabac_state = state + abs_off[cat];
    const int levelgt1_ctx = b_chroma && dc ? 8 : 9;
    dctcoef orig[64], quant[64];
    auto zz = [&](int i) { return cat == 5 ? (int)x264o_zigzag8[i] : cat == 3 ? i : (int)x264o_zigzag4[i]; };
    auto unq = [&](int i) { return dc ? (x264o_unquant4(qp, 0) << 1) : cat == 5 ? x264o_unquant8(qp, zz(i)) : x264o_unquant4(qp, zz(i)); };
    auto wgt = [&](int i) { return dc ? 256u : cat == 5 ? weight8(zz(i)) : weight4(zz(i)); };
    memcpy(orig, dct, sizeof(dctcoef) * num_coefs);
    // the guess: round to nearest (quant_bias0 = (1 << 15) / mf), no dead zone
    {
        int nz = 0;
        for (int i = 0; i < num_coefs; i++) {
            const int m = dc ? mf[0] >> 1 : mf[i], bias = dc ? ((1 << 15) / mf[0]) << 1 : (1 << 15) / mf[i];
            const int c = dct[i];
            dct[i] = (dctcoef)(c > 0 ? ((bias + c) * m) >> 16 : -(((bias - c) * m) >> 16));
            nz |= dct[i];
        }
        if (!nz) return 0;                              // (AC blocks arrive with their DC position cleared)
        for (int i = 0; i < num_coefs; i++) quant[i] = dct[zz(i)];
    }
    int last_nnz = num_coefs - 1;
    while (last_nnz > b_ac && !quant[last_nnz]) last_nnz--;
    // contexts of the significance map: scan position (AC blocks: position among the 15), 8x8: Table 9-43
    auto sigidx = [&](int i) { return cat == 5 ? (int)cabacdec::kSigInc8[i] : i - b_ac; };
    auto lastidx = [&](int i) { return cat == 5 ? (int)cabacdec::kLastInc8[i] : i - b_ac; };

    // a block with nothing but its first coefficient: both candidates priced directly (same result as the search)
    if (last_nnz == 0 && !dc) {
        const int cost_sig = ent(state_sig[0], 1) + ent(state_last[0], 1);
        const int sign_coef = orig[0], q = abs(quant[0]);
        uint64_t bscore = SCORE_MAX;
        int ret = 0;
        for (int abs_level = q - 1; abs_level <= q; abs_level++) {
            const int unquant_abs = (unq(0) * abs_level + 128) >> 8;
            const int d = sign_coef - ((sign_of(unquant_abs, sign_coef) + 8) & ~15);            // DC-only blocks reconstruct in steps of 16
            uint64_t score = (uint64_t)((int64_t)d * d) * wgt(0);
            if (abs_level) {
                unsigned f8 = (unsigned)cost_sig;
                const int prefix = abs_level - 1 < 14 ? abs_level - 1 : 14;
                f8 += ent(cabac_state[1], prefix > 0);
                f8 += tab().size_unary[prefix][cabac_state[5]];
                if (abs_level >= 15) f8 += size_ue_big((unsigned)(abs_level - 15)) << 8;
                score += (uint64_t)f8 * lambda2 >> 4;
            }
            if (score < bscore) { bscore = score; ret = abs_level; }
        }
        dct[0] = (dctcoef)sign_of(ret, sign_coef);
        return dct[0] != 0;
    }

    Trellis T;                                          // (on the stack: the stub device of tests/stub runs several oracle encoders on host threads)
    T.cur = T.nodes[0]; T.prev = T.nodes[1]; T.used = 1; T.lambda2 = lambda2;
    memset(T.nodes, 0, sizeof(T.nodes));
    for (int j = 1; j < 8; j++) T.cur[j].score = SCORE_MAX;
    T.cur[0].score = SCORE_BIAS; T.cur[0].level_idx = 0;
    T.tree[0].abs_level = 0; T.tree[0].next = 0;
    memcpy(T.level_state, cabac_state, 10);
    T.level_state[12] = cabac_state[0]; T.level_state[13] = cabac_state[4]; T.level_state[14] = cabac_state[8]; T.level_state[15] = cabac_state[9];

    int ctx_hi = 0;
    for (int i = last_nnz; i >= b_ac; i--) {
        if (!quant[i]) {
            // nothing to choose; in ctx_lo the all-zero path is spared the significance bit (subtracting from one = adding to the rest)
            if (!ctx_hi) T.cur[0].score -= (uint64_t)ent(state_sig[sigidx(i)], 0) * lambda2 >> 4;
            for (int j = 1; j < (ctx_hi ? 8 : 4); j++) T.set_level(T.cur[j], T.cur[j], 0);
            continue;
        }
        const int sign_coef = orig[zz(i)], i_coef = abs(sign_coef), q = abs(quant[i]);
        { Node *t = T.cur; T.cur = T.prev; T.prev = t; }
        for (int j = ctx_hi; j < 8; j++) T.cur[j].score = SCORE_MAX;
        if (i < num_coefs - 1) {
            T.cost_siglast[0] = ent(state_sig[sigidx(i)], 0);
            const int cost_sig1 = ent(state_sig[sigidx(i)], 1);
            T.cost_siglast[1] = ent(state_last[lastidx(i)], 0) + cost_sig1;
            T.cost_siglast[2] = ent(state_last[lastidx(i)], 1) + cost_sig1;
        } else T.cost_siglast[0] = T.cost_siglast[1] = T.cost_siglast[2] = 0;
        uint64_t ssd0[2], ssd1[2];
        for (int k = 0; k < 2; k++) {
            const int abs_level = q - 1 + k, unquant_abs = (unq(i) * abs_level + 128) >> 8;
            int d = i_coef - unquant_abs;
            ssd1[k] = (uint64_t)((int64_t)d * d) * wgt(i);
            ssd0[k] = ssd1[k];
            if (!i && !dc && !ctx_hi) {
                d = sign_coef - ((sign_of(unquant_abs, sign_coef) + 8) & ~15);
                ssd0[k] = (uint64_t)((int64_t)d * d) * wgt(i);
            }
        }
        if (q == 1) {
            ssd1[0] += (uint64_t)T.cost_siglast[0] * lambda2 >> 4;
            T.coef0(ctx_hi, ssd0[0] - ssd1[0]);
            T.coef1(ctx_hi, ssd0[1] - ssd1[0], ssd1[1] - ssd1[0]);
        } else if (q == 2) {
            T.coef1(ctx_hi, ssd0[0], ssd1[0]);
            T.coefn(ctx_hi, q, ssd0[1], ssd1[1], levelgt1_ctx);
            ctx_hi = 1;
        } else {
            T.coefn(ctx_hi, q - 1, ssd0[0], ssd1[0], levelgt1_ctx);
            T.coefn(ctx_hi, q, ssd0[1], ssd1[1], levelgt1_ctx);
            ctx_hi = 1;
        }
    }
    const Node *bnode = &T.cur[ctx_hi];
    for (int j = ctx_hi + 1; j < (ctx_hi ? 8 : 4); j++) if (T.cur[j].score < bnode->score) bnode = &T.cur[j];
    if (bnode == &T.cur[0]) { memset(dct, 0, sizeof(dctcoef) * num_coefs); return 0; }       // the all-zero path won
    int level = bnode->level_idx, nz = 0;
    for (int i = b_ac; i <= last_nnz; i++) {
        dct[zz(i)] = (dctcoef)sign_of(T.tree[level].abs_level, dct[zz(i)]);
        nz |= dct[zz(i)];
        level = T.tree[level].next;
    }
    return nz != 0;
}

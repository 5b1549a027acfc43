// oracle/cabac_rd.cpp — CABAC as x264's rate-distortion code sees it (TEST INFRASTRUCTURE ONLY; see x264o.h).
//
// x264 ([x264-upstream] encoder/cabac.c, encoder/rdo.c; reached from codec.c:1693) keeps ONE set of context states per slice, moved on by
// the entropy coding of every finished macroblock, and prices a candidate macroblock by running the same syntax over a COPY of those states
// with the arithmetic coder replaced by a bit counter in 1/256 bit units (x264_cabac_size_decision: entropy[state ^ bin], bypass = 256,
// the I16x16 terminate bin = 7).  Two things follow for the checker:
//   * mode 0, "evolve": the finished macroblock's bins in bitstream order, states only — what x264_macroblock_write_cabac leaves behind;
//   * mode 1, "size":   x264_macroblock_size_cabac — no mb_skip_flag, no end_of_slice, and residual blocks walked from the last coefficient
//                       down with significance flags and levels interleaved (cabac_block_residual_*_rd).  For 4x4 / DC blocks every
//                       position has its own context, so the order is immaterial; for 8x8 blocks several positions share one and the
//                       order changes the count, which is why both orders exist here.
// The entropy table is -log2 of the standard's probability model (p_LPS(s) = 0.5 * alpha^s, alpha = (0.01875 / 0.5)^(1/63)) as x264 types
// it: four decimals, then 8.8 fixed point.  Context initialisation and the state transitions come from the checker decoder's own tables
// (cabac_dec.hpp).  Restated from memory like the rest of the oracle: parity unpinned.
#include "cabac_dec.hpp"
#include "x264gpu.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace {

const uint8_t kBx[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
const uint8_t kBy[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
const uint8_t kIdx[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };  // [by][bx]

struct Tables {
    uint16_t entropy[128];        // [sigma * 2 + (bin != mps)]
    Tables()
    {
        const double alpha = pow(0.01875 / 0.5, 1.0 / 63.0);
        for (int s = 0; s < 64; s++) {
            const double p = 0.5 * pow(alpha, s);
            const double mps = floor(-log2(1.0 - p) * 10000.0 + 0.5) / 10000.0, lps = floor(-log2(p) * 10000.0 + 0.5) / 10000.0;
            entropy[2 * s] = (uint16_t)(int)(mps * 256.0 + 0.5);
            entropy[2 * s + 1] = (uint16_t)(int)(lps * 256.0 + 0.5);
        }
    }
};
const Tables &tables() { static const Tables t; return t; }

inline bool intra_type(int t) { return t == X264GPU_MB_I4x4 || t == X264GPU_MB_I8x8 || t == X264GPU_MB_I16x16; }
inline bool skip_type(int t) { return t == X264GPU_MB_P_SKIP || t == X264GPU_MB_B_SKIP; }
inline bool b_type(int t) { return t >= X264GPU_MB_B_DIRECT && t <= X264GPU_MB_B_8x8; }
// list use of 8x8 block k of a B macroblock: 0 list 0, 1 list 1, 2 both, 3 direct
inline int b_use(const x264gpu_mb &m, int k)
{
    if (m.type == X264GPU_MB_B_DIRECT || m.type == X264GPU_MB_B_SKIP || (m.type == X264GPU_MB_B_8x8 && (m.direct8 >> k & 1))) return 3;
    return m.ref[k] >= 0 ? (m.ref1[k] >= 0 ? 2 : 0) : 1;
}

}  // namespace

extern "C" {

typedef struct x264o_cabac_ctx {
    const x264gpu_mb *mbs;
    const int16_t *levels;
    int mbw, mbh, first_row;          /* the slice starts at macroblock row first_row: nothing above it is a neighbour */
    int pslice, num_ref, t8mode;
    uint8_t *amvd;                    /* [macroblock][8x8 block][x, y]: |mvd| capped as x264 keeps it */
    uint8_t *state;                   /* 460 context variables, (pStateIdx << 1) | valMPS */
    int last_dqp, last_qp;            /* mb_qp_delta of the previous macroblock in coding order; QP_Y the entropy coder predicts from */
    int bslice, num_ref1;             /* B slice; active references of list 1 */
    uint8_t *amvd1;                   /* list 1's |mvd| */
} x264o_cabac_ctx;

const uint16_t *x264o_cabac_entropy(void) { return tables().entropy; }

void x264o_cabac_init_states(uint8_t *state, int pslice, int qp)
{
    cabacdec::Engine e;
    e.start(nullptr, 0, 0, pslice != 0, qp);
    for (int i = 0; i < 460; i++) state[i] = (uint8_t)((e.st[i] << 1) | e.mps[i]);
}

}  // extern "C"

namespace {

struct Coder {
    x264o_cabac_ctx &c;
    const bool rd;
    long f8 = 0;
    int cur = 0, done8 = 0;
    struct Nb { bool avail; int ref, mvx, mvy; } cur8[4];
    int lst = 0;                      // the list the motion helpers below read (B slices: 0 / 1)

    Coder(x264o_cabac_ctx &ctx, bool size_mode) : c(ctx), rd(size_mode) {}

    void decision(int ctx, int bin)
    {
        const int st = c.state[ctx], s = st >> 1, mps = st & 1;
        f8 += tables().entropy[2 * s + (bin != mps)];
        if (bin != mps) c.state[ctx] = (uint8_t)((cabacdec::kNextLps[s] << 1) | (s == 0 ? !mps : mps));
        else c.state[ctx] = (uint8_t)(((s < 62 ? s + 1 : 62) << 1) | mps);
    }
    void bypass() { f8 += 256; }
    void ue_bypass(int k, int v)
    {
        while (v >= (1 << k)) { bypass(); v -= 1 << k; k++; }
        bypass();
        while (k--) bypass();
    }

    const x264gpu_mb *left(int mbx, int mby) const { return mbx > 0 ? &c.mbs[mby * c.mbw + mbx - 1] : nullptr; }
    const x264gpu_mb *top(int mbx, int mby) const { return mby > c.first_row ? &c.mbs[(mby - 1) * c.mbw + mbx] : nullptr; }

    Nb block8(int gx, int gy) const
    {
        Nb n = { false, -1, 0, 0 };
        if (gx < 0 || gy < 2 * c.first_row || gx >= 2 * c.mbw || gy >= 2 * c.mbh) return n;
        const int i = (gy >> 1) * c.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i == cur) { if (done8 >> k & 1) return cur8[k]; return n; }
        if (i > cur) return n;
        n.avail = true;
        const x264gpu_mb &m = c.mbs[i];
        if (!intra_type(m.type)) {
            if (lst) { n.ref = m.ref1[k]; if (n.ref >= 0) { n.mvx = m.mv1[k][0]; n.mvy = m.mv1[k][1]; } else n.ref = -1; }
            else { n.ref = m.ref[k]; if (n.ref >= 0) { n.mvx = m.mv[k][0]; n.mvy = m.mv[k][1]; } else n.ref = -1; }
        }
        return n;
    }
    // 8.4.1.3 with the directional rules of the two-partition shapes
    void predict(int mbx, int mby, int bx8, int by8, int w8, int shape, int part, int ref, int &px, int &py) const
    {
        const int gx = 2 * mbx + bx8, gy = 2 * mby + by8;
        Nb a = block8(gx - 1, gy), b = block8(gx, gy - 1), d = block8(gx + w8, gy - 1);
        if (!d.avail) d = block8(gx - 1, gy - 1);
        if (shape == 1) {
            if (part == 0 && b.ref == ref) { px = b.mvx; py = b.mvy; return; }
            if (part == 1 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
        } else if (shape == 2) {
            if (part == 0 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
            if (part == 1 && d.ref == ref) { px = d.mvx; py = d.mvy; return; }
        }
        if (!b.avail && !d.avail && a.avail) { b = a; d = a; }
        const int na = a.ref == ref, nb = b.ref == ref, nd = d.ref == ref;
        if (na + nb + nd == 1) { const Nb &s = na ? a : nb ? b : d; px = s.mvx; py = s.mvy; return; }
        auto med = [](int x, int y, int z) { const int mn = x < y ? x : y, mx = x < y ? y : x; return z < mn ? mn : z > mx ? mx : z; };
        px = med(a.mvx, b.mvx, d.mvx); py = med(a.mvy, b.mvy, d.mvy);
    }
    int amvd_at(int gx, int gy, int comp) const
    {
        if (gx < 0 || gy < 2 * c.first_row || gx >= 2 * c.mbw || gy >= 2 * c.mbh) return 0;
        const int i = (gy >> 1) * c.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i > cur || (i == cur && !(done8 >> k & 1))) return 0;
        return (lst ? c.amvd1 : c.amvd)[((size_t)i * 4 + k) * 2 + comp];
    }
    int ref_gt0_at(int gx, int gy) const
    {
        const Nb n = block8(gx, gy);
        if (!n.avail || n.ref <= 0) return 0;
        const int i = (gy >> 1) * c.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i == cur) return !(cur_direct >> k & 1);
        if (b_type(c.mbs[i].type)) return b_use(c.mbs[i], k) != 3;      // predicted by direct inference: refIdxZeroFlag does not look at it
        return c.mbs[i].type != X264GPU_MB_P_SKIP;
    }
    int cur_direct = 0;               // direct 8x8 blocks of the macroblock being coded
    int pred_intra_mode(int mbx, int mby, int blk) const
    {
        const int bx = kBx[blk], by = kBy[blk];
        const x264gpu_mb &m = c.mbs[cur];
        int ma, mb;
        if (bx > 0) ma = m.i4_mode[kIdx[by][bx - 1]];
        else if (const x264gpu_mb *n = left(mbx, mby)) ma = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[kIdx[by][3]] : 2;
        else return 2;
        if (by > 0) mb = m.i4_mode[kIdx[by - 1][bx]];
        else if (const x264gpu_mb *n = top(mbx, mby)) mb = (n->type == X264GPU_MB_I4x4 || n->type == X264GPU_MB_I8x8) ? n->i4_mode[kIdx[3][bx]] : 2;
        else return 2;
        return ma < mb ? ma : mb;
    }

    // coded_block_flag neighbour terms (9.3.3.1.1.9)
    static int luma_cbf_of(const x264gpu_mb &m, int bx, int by)
    {
        if (m.type == X264GPU_MB_P_SKIP) return 0;
        if (!(m.cbp_luma >> ((by >> 1) * 2 + (bx >> 1)) & 1)) return 0;
        if (m.transform8x8) return 1;
        return (m.nnz >> kIdx[by][bx]) & 1;
    }
    // RD refinement (subme >= 8) prices PARTS of a macroblock: the blocks around a part inside the macroblock then read what x264's
    // non_zero_count cache holds for them — whatever the last encode of any kind left there — and not the record being built
    const uint8_t *nnz_over = nullptr;          // 24 flags: luma blocks 0..15, chroma AC plane * 4 + block
    int cbf_luma(int mbx, int mby, const x264gpu_mb &m, int blk) const
    {
        const int bx = kBx[blk], by = kBy[blk], un = intra_type(m.type) ? 1 : 0;
        int a, b;
        if (bx > 0) a = nnz_over ? nnz_over[kIdx[by][bx - 1]] : luma_cbf_of(m, bx - 1, by); else { const x264gpu_mb *n = left(mbx, mby); a = n ? luma_cbf_of(*n, 3, by) : un; }
        if (by > 0) b = nnz_over ? nnz_over[kIdx[by - 1][bx]] : luma_cbf_of(m, bx, by - 1); else { const x264gpu_mb *n = top(mbx, mby); b = n ? luma_cbf_of(*n, bx, 3) : un; }
        return a + 2 * b;
    }
    int cbf_dc(int mbx, int mby, const x264gpu_mb &m, int bit) const
    {
        const int un = intra_type(m.type) ? 1 : 0;
        auto of = [&](const x264gpu_mb *n) {
            if (!n) return un;
            if (n->type == X264GPU_MB_P_SKIP) return 0;
            if (bit == 24) return n->type == X264GPU_MB_I16x16 ? (int)((n->nnz >> 24) & 1) : 0;
            return n->cbp_chroma ? (int)((n->nnz >> bit) & 1) : 0;
        };
        return of(left(mbx, mby)) + 2 * of(top(mbx, mby));
    }
    int cbf_chroma_ac(int mbx, int mby, const x264gpu_mb &m, int pl, int i) const
    {
        const int bx = i & 1, by = i >> 1, un = intra_type(m.type) ? 1 : 0;
        auto of = [&](const x264gpu_mb &n, int x, int y) { return n.type != X264GPU_MB_P_SKIP && n.cbp_chroma == 2 ? (int)((n.nnz >> (16 + pl * 4 + y * 2 + x)) & 1) : 0; };
        int a, b;
        if (bx > 0) a = nnz_over ? nnz_over[16 + pl * 4 + by * 2] : of(m, 0, by); else { const x264gpu_mb *n = left(mbx, mby); a = n ? of(*n, 1, by) : un; }
        if (by > 0) b = nnz_over ? nnz_over[16 + pl * 4 + bx] : of(m, bx, 0); else { const x264gpu_mb *n = top(mbx, mby); b = n ? of(*n, bx, 1) : un; }
        return a + 2 * b;
    }

    // ---- residual_block_cabac.  cat: 0 luma DC, 1 luma AC, 2 luma 4x4, 3 chroma DC, 4 chroma AC, 5 luma 8x8 ----
    void level(int cat, int v, int &node)
    {
        static const int abs_off[6] = { 227, 237, 247, 257, 266, 426 };
        static const uint8_t lvl1_ctx[8] = { 1, 2, 3, 4, 0, 0, 0, 0 }, gt1_ctx[8] = { 5, 5, 5, 5, 6, 7, 8, 9 };
        static const uint8_t trans[2][8] = { { 1, 2, 3, 3, 4, 5, 6, 7 }, { 4, 4, 4, 4, 5, 6, 7, 7 } };
        const int a = abs(v), ctx = abs_off[cat] + lvl1_ctx[node];
        if (a > 1) {
            decision(ctx, 1);
            const int c2 = abs_off[cat] + gt1_ctx[node];
            for (int i = (a < 15 ? a : 15) - 2; i > 0; i--) decision(c2, 1);
            if (a < 15) decision(c2, 0); else ue_bypass(0, a - 15);
            node = trans[1][node];
        } else { decision(ctx, 0); node = trans[0][node]; }
        bypass();                                      // sign
    }
    void residual(const int16_t *l, int cat)
    {
        static const int sig_off[6] = { 105, 120, 134, 149, 152, 402 }, last_off[6] = { 166, 181, 195, 210, 213, 417 };
        static const int count_m1[6] = { 15, 14, 15, 3, 14, 63 };
        const int n1 = count_m1[cat];
        int last = n1;
        while (last > 0 && !l[last]) last--;
        auto so = [&](int i) { return sig_off[cat] + (cat == 5 ? cabacdec::kSigInc8[i] : i); };
        auto lo = [&](int i) { return last_off[cat] + (cat == 5 ? cabacdec::kLastInc8[i] : i); };
        int node = 0;
        if (rd) {
            // x264's size-only walk: from the last coefficient down, flags and level of each position together
            if (last != n1) { decision(so(last), 1); decision(lo(last), 1); }
            level(cat, l[last], node);
            for (int i = last - 1; i >= 0; i--) {
                if (l[i]) { decision(so(i), 1); decision(lo(i), 0); level(cat, l[i], node); }
                else decision(so(i), 0);
            }
        } else {
            for (int i = 0; i < last; i++) { decision(so(i), l[i] != 0); if (l[i]) decision(lo(i), 0); }
            if (last != n1) { decision(so(last), 1); decision(lo(last), 1); }
            for (int i = last; i >= 0; i--) if (l[i]) level(cat, l[i], node);
        }
    }
    void block_cbf(const int16_t *l, int n, int cat, int inc)
    {
        static const int cbf_off[5] = { 85, 89, 93, 97, 101 };
        int nz = 0;
        for (int i = 0; i < n; i++) nz |= l[i];
        decision(cbf_off[cat] + inc, nz != 0);
        if (nz) residual(l, cat);
    }

    void mb_type_intra(const x264gpu_mb &m, int c0, int c1, int c2, int c3, int c4, int c5)
    {
        if (m.type != X264GPU_MB_I16x16) { decision(c0, 0); return; }
        decision(c0, 1);
        if (rd) f8 += 7;                               // the terminate bin (not I_PCM) as x264's size macro prices it; no state behind it
        decision(c1, m.cbp_luma != 0);
        if (!m.cbp_chroma) decision(c2, 0);
        else { decision(c2, 1); decision(c3, m.cbp_chroma >> 1); }
        decision(c4, m.i16_mode >> 1);
        decision(c5, m.i16_mode & 1);
    }
    void mvd(int mbx, int mby, int b8, int w8, int h8, int comp, int val)
    {
        const int gx = 2 * mbx + (b8 & 1), gy = 2 * mby + (b8 >> 1);
        const int sum = amvd_at(gx - 1, gy, comp) + amvd_at(gx, gy - 1, comp);
        const int base = comp ? 47 : 40, inc = (sum > 2) + (sum > 32), a = abs(val);
        static const uint8_t ctxes[8] = { 3, 4, 5, 6, 6, 6, 6, 6 };
        if (!a) decision(base + inc, 0);
        else {
            decision(base + inc, 1);
            for (int i = 1; i < (a < 9 ? a : 9); i++) decision(base + ctxes[i - 1], 1);
            if (a < 9) decision(base + ctxes[a - 1], 0); else ue_bypass(3, a - 9);
            bypass();
        }
        const uint8_t capped = (uint8_t)(a < 66 ? a : 66);
        for (int y = b8 >> 1; y < (b8 >> 1) + h8; y++)
            for (int x = b8 & 1; x < (b8 & 1) + w8; x++) (lst ? c.amvd1 : c.amvd)[((size_t)cur * 4 + y * 2 + x) * 2 + comp] = capped;
    }
    void ref_idx(int mbx, int mby, int b8, int ref)
    {
        const int gx = 2 * mbx + (b8 & 1), gy = 2 * mby + (b8 >> 1);
        int ctx = ref_gt0_at(gx - 1, gy) + 2 * ref_gt0_at(gx, gy - 1);
        for (int r = ref; r > 0; r--) { decision(54 + ctx, 1); ctx = (ctx >> 2) + 4; }
        decision(54 + ctx, 0);
    }

    // mb_type of a B slice (Table 9-37 b; ctxIdx 27..35): the bin string of the value, context 27 + {0..2} for bin 0, 27 + 3 for bin 1,
    // 27 + 5 - b1 for bin 2, 27 + 5 for the rest
    void mb_type_b(int value, int ctx0)
    {
        static const char *const bins[24] = { "0", "100", "101", "110000", "110001", "110010", "110011", "110100", "110101", "110110", "110111", "111110",
                                              "1110000", "1110001", "1110010", "1110011", "1110100", "1110101", "1110110", "1110111", "1111000", "1111001",
                                              "111111", "111101" /* prefix of the intra types */ };
        const char *b = bins[value];
        for (int i = 0; b[i]; i++) decision(i == 0 ? 27 + ctx0 : i == 1 ? 27 + 3 : i == 2 ? 27 + 5 - (b[1] - '0') : 27 + 5, b[i] - '0');
    }
    void sub_mb_type_b(int use)          // 0 list 0, 1 list 1, 2 both, 3 direct (Table 9-37: B_Direct_8x8 "0", L0 "100", L1 "101", Bi "11000"; ctxIdx 36..39)
    {
        if (use == 3) { decision(36, 0); return; }
        decision(36, 1);
        if (use == 2) { decision(37, 1); decision(38, 0); decision(39, 0); decision(39, 0); return; }
        decision(37, 0);
        decision(39, use == 1);
    }
    void macroblock_b(int mbx, int mby, const x264gpu_mb &m)
    {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        const x264gpu_mb *L = left(mbx, mby), *T = top(mbx, mby);
        const int ctx0 = (L && L->type != X264GPU_MB_B_SKIP && L->type != X264GPU_MB_B_DIRECT) + (T && T->type != X264GPU_MB_B_SKIP && T->type != X264GPU_MB_B_DIRECT);
        cur_direct = 0;
        if (m.type == X264GPU_MB_B_DIRECT) { mb_type_b(0, ctx0); cur_direct = 15; return; }
        const int part = m.partition & 3, nparts = part == 0 ? 1 : part == 3 ? 4 : 2;
        int use[4];
        for (int k = 0; k < nparts; k++) use[k] = b_use(m, geom[part][k][1] * 2 + geom[part][k][0]);
        if (part == 3) {
            mb_type_b(22, ctx0);
            for (int k = 0; k < 4; k++) { sub_mb_type_b(use[k]); if (use[k] == 3) cur_direct |= 1 << k; }
        } else if (part == 0) mb_type_b(1 + use[0], ctx0);
        else {
            // Table 7-14: 4 + 2 * pair + (8x16), pairs in the order L0_L0, L1_L1, L0_L1, L1_L0, L0_Bi, L1_Bi, Bi_L0, Bi_L1, Bi_Bi
            static const int8_t pair_of[3][3] = { { 0, 2, 4 }, { 3, 1, 5 }, { 6, 7, 8 } };
            mb_type_b(4 + 2 * pair_of[use[0]][use[1]] + (part == 2), ctx0);
        }
        for (lst = 0; lst < 2; lst++) {
            if ((lst ? c.num_ref1 : c.num_ref) <= 1) continue;
            done8 = 0;
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[part][k];
                const int b8 = g[1] * 2 + g[0];
                if (use[k] == 3 || use[k] == 1 - lst) {
                    // the block does not send this list's index: for the neighbour rules it has none (direct: never "greater than 0")
                    for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) { cur8[y * 2 + x] = Nb{ true, use[k] == 3 ? (lst ? m.ref1[b8] : m.ref[b8]) : -1, 0, 0 }; done8 |= 1 << (y * 2 + x); }
                    continue;
                }
                const int r = lst ? m.ref1[b8] : m.ref[b8];
                ref_idx(mbx, mby, b8, r);
                for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) { cur8[y * 2 + x] = Nb{ true, r, 0, 0 }; done8 |= 1 << (y * 2 + x); }
            }
        }
        for (lst = 0; lst < 2; lst++) {
            done8 = 0;
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[part][k];
                const int b8 = g[1] * 2 + g[0];
                const int r = lst ? m.ref1[b8] : m.ref[b8], vx = lst ? m.mv1[b8][0] : m.mv[b8][0], vy = lst ? m.mv1[b8][1] : m.mv[b8][1];
                if (!(use[k] == 3 || use[k] == 1 - lst)) {
                    int px, py;
                    predict(mbx, mby, g[0], g[1], g[2], part, k, r, px, py);
                    mvd(mbx, mby, b8, g[2], g[3], 0, vx - px);
                    mvd(mbx, mby, b8, g[2], g[3], 1, vy - py);
                }
                for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) { cur8[y * 2 + x] = Nb{ true, r >= 0 ? r : -1, r >= 0 ? vx : 0, r >= 0 ? vy : 0 }; done8 |= 1 << (y * 2 + x); }
            }
        }
        lst = 0;
    }

    void macroblock(int mbx, int mby)
    {
        cur = mby * c.mbw + mbx; done8 = 0; lst = 0; cur_direct = 0;
        const x264gpu_mb &m = c.mbs[cur];
        const int16_t *lv = c.levels + (size_t)cur * X264GPU_MB_LEVELS;
        const x264gpu_mb *L = left(mbx, mby), *T = top(mbx, mby);
        memset(c.amvd + (size_t)cur * 8, 0, 8);
        if (c.bslice) memset(c.amvd1 + (size_t)cur * 8, 0, 8);
        if (c.bslice && !rd) {
            decision(24 + (L && !skip_type(L->type)) + (T && !skip_type(T->type)), m.type == X264GPU_MB_B_SKIP);
            if (m.type == X264GPU_MB_B_SKIP) { c.last_dqp = 0; return; }
        }
        if (c.pslice && !c.bslice && !rd) {
            decision(11 + (L && L->type != X264GPU_MB_P_SKIP) + (T && T->type != X264GPU_MB_P_SKIP), m.type == X264GPU_MB_P_SKIP);
            if (m.type == X264GPU_MB_P_SKIP) { c.last_dqp = 0; return; }
        }
        const bool intra = intra_type(m.type);
        if (c.bslice) {
            if (intra) {
                const int ctx0 = (L && L->type != X264GPU_MB_B_SKIP && L->type != X264GPU_MB_B_DIRECT) + (T && T->type != X264GPU_MB_B_SKIP && T->type != X264GPU_MB_B_DIRECT);
                mb_type_b(23, ctx0);
                mb_type_intra(m, 32, 32 + 1, 32 + 2, 32 + 2, 32 + 3, 32 + 3);
            } else macroblock_b(mbx, mby, m);
        } else if (!c.pslice) {
            const int ctx = (L && L->type != X264GPU_MB_I4x4 && L->type != X264GPU_MB_I8x8) + (T && T->type != X264GPU_MB_I4x4 && T->type != X264GPU_MB_I8x8);
            mb_type_intra(m, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
        } else if (intra) { decision(14, 1); mb_type_intra(m, 17, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3); }
        else if (m.partition == 3) { decision(14, 0); decision(15, 0); decision(16, 1); }
        else {
            decision(14, 0);
            if (m.partition == 0) { decision(15, 0); decision(16, 0); }
            else { decision(15, 1); decision(17, m.partition == 1); }
        }
        const int t8ctx = 399 + (L && L->transform8x8) + (T && T->transform8x8);
        if (intra) {
            if (m.type != X264GPU_MB_I16x16) {
                if (c.t8mode) decision(t8ctx, m.type == X264GPU_MB_I8x8);
                for (int b = 0; b < 16; b += m.type == X264GPU_MB_I8x8 ? 4 : 1) {
                    const int pm = pred_intra_mode(mbx, mby, b);
                    int mode = m.i4_mode[b];
                    if (mode == pm) decision(68, 1);
                    else {
                        decision(68, 0);
                        if (mode > pm) mode--;
                        decision(69, mode & 1); decision(69, (mode >> 1) & 1); decision(69, mode >> 2);
                    }
                }
            }
            const int ctx = (L && intra_type(L->type) && L->chroma_mode != 0) + (T && intra_type(T->type) && T->chroma_mode != 0);
            if (!m.chroma_mode) decision(64 + ctx, 0);
            else { decision(64 + ctx, 1); decision(64 + 3, m.chroma_mode > 1); if (m.chroma_mode > 1) decision(64 + 3, m.chroma_mode > 2); }
        } else if (!c.bslice) {
            static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                                  { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
            const int nparts = m.partition == 0 ? 1 : m.partition == 3 ? 4 : 2;
            if (m.partition == 3) for (int k = 0; k < 4; k++) decision(21, 1);
            if (c.num_ref > 1)
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[m.partition][k];
                    const int b8 = g[1] * 2 + g[0];
                    ref_idx(mbx, mby, b8, m.ref[b8]);
                    for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) { cur8[y * 2 + x] = Nb{ true, m.ref[b8], 0, 0 }; done8 |= 1 << (y * 2 + x); }
                }
            done8 = 0;
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[m.partition][k];
                const int b8 = g[1] * 2 + g[0];
                int px, py;
                predict(mbx, mby, g[0], g[1], g[2], m.partition, k, m.ref[b8], px, py);
                mvd(mbx, mby, b8, g[2], g[3], 0, m.mv[b8][0] - px);
                mvd(mbx, mby, b8, g[2], g[3], 1, m.mv[b8][1] - py);
                for (int y = g[1]; y < g[1] + g[3]; y++) for (int x = g[0]; x < g[0] + g[2]; x++) { cur8[y * 2 + x] = Nb{ true, m.ref[b8], m.mv[b8][0], m.mv[b8][1] }; done8 |= 1 << (y * 2 + x); }
            }
        }
        if (m.type != X264GPU_MB_I16x16) {
            for (int b8 = 0; b8 < 4; b8++) {
                const int x = b8 & 1, y = b8 >> 1;
                const int a = x ? !((m.cbp_luma >> (b8 - 1)) & 1) : L ? !((L->cbp_luma >> (b8 + 1)) & 1) : 0;
                const int b = y ? !((m.cbp_luma >> (b8 - 2)) & 1) : T ? !((T->cbp_luma >> (b8 + 2)) & 1) : 0;
                decision(73 + a + 2 * b, (m.cbp_luma >> b8) & 1);
            }
            decision(77 + (L && L->cbp_chroma) + 2 * (T && T->cbp_chroma), m.cbp_chroma != 0);
            if (m.cbp_chroma) decision(77 + 4 + (L && L->cbp_chroma == 2) + 2 * (T && T->cbp_chroma == 2), m.cbp_chroma == 2);
        }
        if (!intra && c.t8mode && m.cbp_luma) decision(t8ctx, m.transform8x8);
        if (m.cbp_luma || m.cbp_chroma || m.type == X264GPU_MB_I16x16) {
            int dqp = (int)m.qp - c.last_qp;
            // an I16x16 with nothing coded, DC included, never raises the quantiser (x264's qp_delta writers): it is sent as "no change"
            if (m.type == X264GPU_MB_I16x16 && !m.cbp_luma && !m.cbp_chroma && !((m.nnz >> 24) & 1) && dqp > 0) dqp = 0;
            int ctx = c.last_dqp != 0;
            if (dqp) {
                if (dqp < -26) dqp += 52; else if (dqp > 25) dqp -= 52;
                int val = dqp > 0 ? 2 * dqp - 1 : -2 * dqp;
                do { decision(60 + ctx, 1); ctx = 2 + (ctx >> 1); } while (--val);
            }
            decision(60 + ctx, 0);
            if (!rd) c.last_dqp = dqp;
            if (m.type == X264GPU_MB_I16x16) {
                block_cbf(lv + X264GPU_LV_LUMA_DC, 16, 0, cbf_dc(mbx, mby, m, 24));
                if (m.cbp_luma) for (int b = 0; b < 16; b++) block_cbf(lv + b * 16 + 1, 15, 1, cbf_luma(mbx, mby, m, b));
            } else if (m.transform8x8) {
                for (int i8 = 0; i8 < 4; i8++)
                    if ((m.cbp_luma >> i8) & 1) {
                        int16_t l8[64];
                        for (int z = 0; z < 64; z++) l8[z] = lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)];
                        residual(l8, 5);
                    }
            } else {
                for (int b = 0; b < 16; b++) if ((m.cbp_luma >> (b >> 2)) & 1) block_cbf(lv + b * 16, 16, 2, cbf_luma(mbx, mby, m, b));
            }
            if (m.cbp_chroma) {
                for (int pl = 0; pl < 2; pl++) block_cbf(lv + X264GPU_LV_CHROMA_DC + pl * 4, 4, 3, cbf_dc(mbx, mby, m, 25 + pl));
                if (m.cbp_chroma == 2)
                    for (int pl = 0; pl < 2; pl++)
                        for (int k = 0; k < 4; k++) block_cbf(lv + X264GPU_LV_CHROMA_AC + (pl * 4 + k) * 16 + 1, 15, 4, cbf_chroma_ac(mbx, mby, m, pl, k));
            }
        } else if (!rd) c.last_dqp = 0;
    }

    // ---- the size functions of x264's RD refinement (encoder/rdo.c partition_size_cabac, partition_i4x4 / _i8x8_size_cabac,
    //      chroma_size_cabac): parts of the macroblock that sits in the record, no macroblock type, reference index, cbp (inter) or qp bits ----
    void intra_pred_mode_bins(int mbx, int mby, int b)
    {
        const x264gpu_mb &m = c.mbs[cur];
        const int pm = pred_intra_mode(mbx, mby, b);
        int mode = m.i4_mode[b];
        if (mode == pm) decision(68, 1);
        else {
            decision(68, 0);
            if (mode > pm) mode--;
            decision(69, mode & 1); decision(69, (mode >> 1) & 1); decision(69, mode >> 2);
        }
    }
    // P_L0 16x8 / 8x16 half or P_8x8 block i8 (pixel: 1 16x8, 2 8x16, 3 8x8); done = the 8x8 blocks of the macroblock refined before it
    // (their final vectors in the record, their |mvd| in c.amvd)
    void partition_p(int mbx, int mby, int i8, int pixel, int done)
    {
        cur = mby * c.mbw + mbx; done8 = done; lst = 0; cur_direct = 0;
        const x264gpu_mb &m = c.mbs[cur];
        const int16_t *lv = c.levels + (size_t)cur * X264GPU_MB_LEVELS;
        for (int k = 0; k < 4; k++) cur8[k] = Nb{ true, m.ref[k], m.mv[k][0], m.mv[k][1] };
        const int bx = i8 & 1, by = i8 >> 1, w8 = pixel == 1 ? 2 : 1, h8 = pixel == 2 ? 2 : 1;
        int px, py;
        predict(mbx, mby, bx, by, w8, pixel == 3 ? 3 : pixel, pixel == 1 ? by : bx, m.ref[i8], px, py);
        mvd(mbx, mby, i8, w8, h8, 0, m.mv[i8][0] - px);
        mvd(mbx, mby, i8, w8, h8, 1, m.mv[i8][1] - py);
        if (pixel == 3) decision(21, 1);                       // sub_mb_type P_L0_8x8
        for (int j = pixel < 3; j >= 0; j--) {
            if ((m.cbp_luma >> i8) & 1) {
                if (m.transform8x8) {
                    int16_t l8[64];
                    for (int z = 0; z < 64; z++) l8[z] = lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)];
                    residual(l8, 5);
                } else for (int b = i8 * 4; b < i8 * 4 + 4; b++) block_cbf(lv + b * 16, 16, 2, cbf_luma(mbx, mby, m, b));
            }
            if (m.cbp_chroma)
                for (int pl = 0; pl < 2; pl++) block_cbf(lv + X264GPU_LV_CHROMA_AC + (pl * 4 + i8) * 16 + 1, 15, 4, cbf_chroma_ac(mbx, mby, m, pl, i8));
            i8 += pixel == 1 ? 1 : 2;
        }
    }
    // a part of a B macroblock (x264_partition_size_cabac, B types): the vector differences of the lists the part uses — no sub_mb_type bins —
    // then the residual of its 8x8 blocks.  The record holds the whole macroblock; done = the 8x8 blocks refined before this part
    void partition_b(int mbx, int mby, int i8, int pixel, int done)
    {
        cur = mby * c.mbw + mbx; cur_direct = 0;
        const x264gpu_mb &m = c.mbs[cur];
        const int16_t *lv = c.levels + (size_t)cur * X264GPU_MB_LEVELS;
        const int bx = i8 & 1, by = i8 >> 1, w8 = pixel == 1 ? 2 : 1, h8 = pixel == 2 ? 2 : 1;
        const int use = b_use(m, i8);
        for (lst = 0; lst < 2; lst++) {
            if (use == 3 || use == 1 - lst) continue;
            done8 = done;
            for (int k = 0; k < 4; k++) { const int r = lst ? m.ref1[k] : m.ref[k]; cur8[k] = Nb{ true, r >= 0 ? r : -1, r >= 0 ? (lst ? m.mv1[k][0] : m.mv[k][0]) : 0, r >= 0 ? (lst ? m.mv1[k][1] : m.mv[k][1]) : 0 }; }
            const int r = lst ? m.ref1[i8] : m.ref[i8], vx = lst ? m.mv1[i8][0] : m.mv[i8][0], vy = lst ? m.mv1[i8][1] : m.mv[i8][1];
            int px, py;
            predict(mbx, mby, bx, by, w8, pixel == 3 ? 3 : pixel, pixel == 1 ? by : bx, r, px, py);
            mvd(mbx, mby, i8, w8, h8, 0, vx - px);
            mvd(mbx, mby, i8, w8, h8, 1, vy - py);
        }
        lst = 0;
        for (int j = pixel < 3; j >= 0; j--) {
            if ((m.cbp_luma >> i8) & 1) {
                if (m.transform8x8) {
                    int16_t l8[64];
                    for (int z = 0; z < 64; z++) l8[z] = lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)];
                    residual(l8, 5);
                } else for (int b = i8 * 4; b < i8 * 4 + 4; b++) block_cbf(lv + b * 16, 16, 2, cbf_luma(mbx, mby, m, b));
            }
            if (m.cbp_chroma)
                for (int pl = 0; pl < 2; pl++) block_cbf(lv + X264GPU_LV_CHROMA_AC + (pl * 4 + i8) * 16 + 1, 15, 4, cbf_chroma_ac(mbx, mby, m, pl, i8));
            i8 += pixel == 1 ? 1 : 2;
        }
    }
    void part_i4x4(int mbx, int mby, int idx)
    {
        cur = mby * c.mbw + mbx;
        const x264gpu_mb &m = c.mbs[cur];
        intra_pred_mode_bins(mbx, mby, idx);
        block_cbf(c.levels + (size_t)cur * X264GPU_MB_LEVELS + idx * 16, 16, 2, cbf_luma(mbx, mby, m, idx));
    }
    void part_i8x8(int mbx, int mby, int i8)
    {
        cur = mby * c.mbw + mbx;
        const x264gpu_mb &m = c.mbs[cur];
        const int16_t *lv = c.levels + (size_t)cur * X264GPU_MB_LEVELS;
        const x264gpu_mb *L = left(mbx, mby), *T = top(mbx, mby);
        intra_pred_mode_bins(mbx, mby, i8 * 4);
        for (int b8 = 0; b8 < 4; b8++) {                       // cabac_cbp_luma: all four bits
            const int x = b8 & 1, y = b8 >> 1;
            const int a = x ? !((m.cbp_luma >> (b8 - 1)) & 1) : L ? !((L->cbp_luma >> (b8 + 1)) & 1) : 0;
            const int b = y ? !((m.cbp_luma >> (b8 - 2)) & 1) : T ? !((T->cbp_luma >> (b8 + 2)) & 1) : 0;
            decision(73 + a + 2 * b, (m.cbp_luma >> b8) & 1);
        }
        if ((m.cbp_luma >> i8) & 1) {
            int16_t l8[64];
            for (int z = 0; z < 64; z++) l8[z] = lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)];
            residual(l8, 5);
        }
    }
    void chroma_part(int mbx, int mby)
    {
        cur = mby * c.mbw + mbx;
        const x264gpu_mb &m = c.mbs[cur];
        const int16_t *lv = c.levels + (size_t)cur * X264GPU_MB_LEVELS;
        const x264gpu_mb *L = left(mbx, mby), *T = top(mbx, mby);
        const int ctx = (L && intra_type(L->type) && L->chroma_mode != 0) + (T && intra_type(T->type) && T->chroma_mode != 0);
        if (!m.chroma_mode) decision(64 + ctx, 0);
        else { decision(64 + ctx, 1); decision(64 + 3, m.chroma_mode > 1); if (m.chroma_mode > 1) decision(64 + 3, m.chroma_mode > 2); }
        decision(77 + (L && L->cbp_chroma) + 2 * (T && T->cbp_chroma), m.cbp_chroma != 0);
        if (m.cbp_chroma) {
            decision(77 + 4 + (L && L->cbp_chroma == 2) + 2 * (T && T->cbp_chroma == 2), m.cbp_chroma == 2);
            for (int pl = 0; pl < 2; pl++) block_cbf(lv + X264GPU_LV_CHROMA_DC + pl * 4, 4, 3, cbf_dc(mbx, mby, m, 25 + pl));
            if (m.cbp_chroma == 2)
                for (int pl = 0; pl < 2; pl++)
                    for (int k = 0; k < 4; k++) block_cbf(lv + X264GPU_LV_CHROMA_AC + (pl * 4 + k) * 16 + 1, 15, 4, cbf_chroma_ac(mbx, mby, m, pl, k));
        }
    }
};

}  // namespace

// The part sizes of RD refinement (x264 subme >= 8), 1/256 bit; c->state is a scratch copy of the slice's context variables; nnzc = x264's
// non_zero_count cache of the macroblock (24 flags) as the encodes before left it.  kind: 0 inter part (a = i8, b = pixel, d = done mask),
// 1 Intra_4x4 block a, 2 Intra_8x8 block a, 3 the chroma of an intra macroblock
extern "C" long x264o_cabac_part(x264o_cabac_ctx *c, int mbx, int mby, int kind, int a, int b, int d, const uint8_t *nnzc)
{
    Coder k(*c, true);
    k.nnz_over = nnzc;
    if (kind == 0) k.partition_p(mbx, mby, a, b, d);
    else if (kind == 1) k.part_i4x4(mbx, mby, a);
    else if (kind == 2) k.part_i8x8(mbx, mby, a);
    else if (kind == 4) k.partition_b(mbx, mby, a, b, d);
    else k.chroma_part(mbx, mby);
    return k.f8;
}

// mode 0: move c->state (and last_dqp, the macroblock's |mvd| entries) past the finished macroblock; mode 1: price the candidate that sits
// in the macroblock's record and levels — c->state must then be a scratch copy.  Returns the count in 1/256 bit units.
extern "C" long x264o_cabac_mb(x264o_cabac_ctx *c, int mbx, int mby, int size_mode)
{
    Coder k(*c, size_mode != 0);
    k.macroblock(mbx, mby);
    return k.f8;
}

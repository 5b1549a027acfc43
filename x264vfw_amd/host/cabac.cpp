// cabac.cpp — host-side CABAC slice data (ITU-T H.264 7.3.4, 7.3.5, 9.3) from the GPU's macroblock records + levels: the entropy coder x264's
// preset medium uses ([x264-upstream] encoder/cabac.c + common/cabac.c behind x264_encoder_encode, reference call site codec.c:1693; north
// star: "CABAC/entropy left on the host").  I and P slices, frame macroblocks, 4:2:0, cabac_init_idc 0, partitions down to 8x8.
// The arithmetic coder is the standard's own formulation (9.3.4.2: PutBit with outstanding bits), bit by bit.
#include "host.hpp"
#include "cabac_tables.hpp"
#include <stdlib.h>
#include <string.h>

namespace x264host {

namespace {

const uint8_t kBlkX[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
const uint8_t kBlkY[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
const uint8_t kIdxOf[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };  // [by][bx]

inline bool is_intra(const x264gpu_mb &m) { return m.type == X264GPU_MB_I4x4 || m.type == X264GPU_MB_I8x8 || m.type == X264GPU_MB_I16x16; }
inline bool is_skip(const x264gpu_mb &m) { return m.type == X264GPU_MB_P_SKIP || m.type == X264GPU_MB_B_SKIP; }
inline bool is_b(const x264gpu_mb &m) { return m.type >= X264GPU_MB_B_DIRECT && m.type <= X264GPU_MB_B_8x8; }
// how 8x8 block k of a B macroblock is predicted: 0 list 0, 1 list 1, 2 both, 3 direct
inline int b_use(const x264gpu_mb &m, int k)
{
    if (m.type == X264GPU_MB_B_DIRECT || m.type == X264GPU_MB_B_SKIP || (m.type == X264GPU_MB_B_8x8 && (m.direct8 >> k & 1))) return 3;
    return m.ref[k] >= 0 ? (m.ref1[k] >= 0 ? 2 : 0) : 1;
}

// The arithmetic coder of 9.3.4.2 with its bits kept in one wide register instead of being put out one by one: `low` holds the 10 bits of codILow and, above them,
// the bits already shifted out but not yet written (`queue` + 8 of them; the first bit of the code word, which 9.3.4.2 drops, and a carry sit above those).  A byte leaves
// when eight are there; a byte of 0xff is held back (`ff_run`) until the byte behind it shows whether a carry still comes — the outstanding bits of PutBit, byte-wise.
// Same code words as the bit-by-bit formulation of the round before (tools/entropy_bench.cpp: same checksums on the device's records), ~4 x its speed.
// (pStateIdx << 1 | valMPS) after a bin: [state byte << 1 | the bin was the less probable symbol]
struct NextState {
    uint8_t t[256];
    NextState()
    {
        for (int v = 0; v < 128; v++) {
            const int s = v >> 1, m = v & 1;
            t[v << 1] = (uint8_t)(((s < 62 ? s + 1 : 62) << 1) | m);
            t[(v << 1) | 1] = (uint8_t)((cabac_trans_lps[s] << 1) | (m ^ (s == 0)));
        }
    }
};
static const NextState kNextState;
static const uint8_t *const kNext = kNextState.t;

struct Cabac {
    std::vector<uint8_t> buf;          // the slice data's bytes so far (behind the slice header)
    uint64_t low = 0;
    uint32_t range = 510;
    int queue = -9;                    // bits above the 10-bit window, minus 8 (and minus the dropped first bit): a byte leaves when it reaches 0
    long ff_run = 0;                   // bytes of 0xff held back
    long head_bits = 0;                // bits of the slice header in front (pos() counts from the start of the RBSP, as the bit writer did)
    uint8_t st[460];                   // (pStateIdx << 1) | valMPS per context

    long pos() const { return head_bits + 8 * ((long)buf.size() + ff_run) + queue + 8; }          // x264_cabac_pos: bits out, the pending ones included
    void init(bool islice, int qp)
    {
        memset(st, 0, sizeof(st));
        auto set = [&](int ctx, const CabacInitRow &r) {
            const int m = islice ? r.mi : r.mp, n = islice ? r.ni : r.np;
            int pre = ((m * (qp < 0 ? 0 : qp > 51 ? 51 : qp)) >> 4) + n;
            pre = pre < 1 ? 1 : pre > 126 ? 126 : pre;
            st[ctx] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
        };
        for (int i = 0; i < 276; i++) set(i, cabac_init_0_275[i]);
        for (int i = 0; i < 37; i++) set(399 + i, cabac_init_399_435[i]);
    }
    __attribute__((always_inline)) inline void put_byte()
    {
        if (queue < 0) return;
        const uint32_t o = (uint32_t)(low >> (queue + 10));          // carry (bit 8) + the byte
        low &= (0x400ull << queue) - 1;
        queue -= 8;
        if ((o & 0xff) == 0xff) { ff_run++; return; }                  // (never with the carry set: a carry leaves a byte of 0x00 behind)
        const uint32_t carry = o >> 8;
        if (!buf.empty()) buf.back() = (uint8_t)(buf.back() + carry);  // (in front of the first byte it is the code word's first bit: dropped, 9.3.4.2 firstBitFlag)
        for (; ff_run > 0; ff_run--) buf.push_back((uint8_t)(0xff + carry));
        buf.push_back((uint8_t)o);
    }
    __attribute__((always_inline)) inline void renorm()
    {
        if (range >= 256) return;
        const int shift = __builtin_clz(range) - 23;
        range <<= shift; low <<= shift; queue += shift;
        put_byte();
    }
    __attribute__((always_inline)) inline void decision(int ctx, int bin)
    {
        // without a branch on the bin (it is the one branch of the coder a predictor cannot learn): both outcomes computed, one selected
        const uint32_t v = st[ctx], rlps = cabac_range_lps[v >> 1][(range >> 6) & 3];
        const uint32_t lps = (uint32_t)bin ^ (v & 1), mask = 0u - lps;          // 1 / all ones: the less probable symbol
        const uint32_t rmps = range - rlps;
        low += rmps & mask;
        range = rmps + ((rlps - rmps) & mask);
        st[ctx] = kNext[(v << 1) | lps];
        const int shift = __builtin_clz(range) - 23;                            // 0 when range >= 256
        range <<= shift; low <<= shift; queue += shift;
        put_byte();
    }
    __attribute__((always_inline)) inline void bypass(int bin)
    {
        low = (low << 1) + (range & (0u - (uint32_t)(bin != 0)));
        queue++;
        put_byte();
    }
    // The same coder on a copy of its registers (the residual walk keeps them in machine registers: through `this` every store to a context byte — a uint8_t, which may alias
    // anything — makes the compiler reload low / range / queue): regs() at the start of a block, set_regs() at its end.
    struct R { uint64_t low; uint32_t range; int queue; };
    R regs() const { return R{ low, range, queue }; }
    void set_regs(const R &r) { low = r.low; range = r.range; queue = r.queue; }
    __attribute__((always_inline)) inline void byte_out(R &r)
    {
        const uint32_t o = (uint32_t)(r.low >> (r.queue + 10));
        r.low &= (0x400ull << r.queue) - 1;
        r.queue -= 8;
        if ((o & 0xff) == 0xff) { ff_run++; return; }
        const uint32_t carry = o >> 8;
        if (!buf.empty()) buf.back() = (uint8_t)(buf.back() + carry);
        for (; ff_run > 0; ff_run--) buf.push_back((uint8_t)(0xff + carry));
        buf.push_back((uint8_t)o);
    }
    __attribute__((always_inline)) inline void decision(R &r, int ctx, int bin)
    {
        const uint32_t v = st[ctx], rlps = cabac_range_lps[v >> 1][(r.range >> 6) & 3];
        const uint32_t lps = (uint32_t)bin ^ (v & 1), mask = 0u - lps;
        const uint32_t rmps = r.range - rlps;
        r.low += rmps & mask;
        r.range = rmps + ((rlps - rmps) & mask);
        st[ctx] = kNext[(v << 1) | lps];
        const int shift = __builtin_clz(r.range) - 23;
        r.range <<= shift; r.low <<= shift; r.queue += shift;
        if (r.queue >= 0) byte_out(r);
    }
    __attribute__((always_inline)) inline void bypass(R &r, int bin)
    {
        r.low = (r.low << 1) + (r.range & (0u - (uint32_t)(bin != 0)));
        r.queue++;
        if (r.queue >= 0) byte_out(r);
    }
    void ue_bypass(R &r, int k, int v)
    {
        while (v >= (1 << k)) { bypass(r, 1); v -= 1 << k; k++; }
        bypass(r, 0);
        while (k--) bypass(r, (v >> k) & 1);
    }
    void ue_bypass(int k, int v)       // Exp-Golomb of order k, bypass bins (9.3.2.3 suffix)
    {
        while (v >= (1 << k)) { bypass(1); v -= 1 << k; k++; }
        bypass(0);
        while (k--) bypass((v >> k) & 1);
    }
    void terminate(int bin)
    {
        range -= 2;
        if (bin) {
            low += range;
            range = 2; renorm();                       // EncodeFlush: seven shifts
            low |= 0x80;                               // ... then bits 9 and 8 of codILow and the rbsp_stop_one_bit
            low <<= 3; queue += 3;
            put_byte();
            if (queue > -8) { const int pad = -queue; low <<= pad; queue += pad; put_byte(); }      // what is left, zeros behind it up to the byte boundary
            for (; ff_run > 0; ff_run--) buf.push_back(0xff);
        } else renorm();
    }
};

struct CabacSlice {
    const SliceParams &p;
    const x264gpu_mb *mbs;
    const int16_t *levels;
    const x264gpu_level_index *index = nullptr;          // the levels are packed (host.hpp mb_levels)
    int16_t lvbuf[X264GPU_MB_LEVELS];
    Cabac &cb;
    std::vector<uint8_t> amvd, amvd1;   // per macroblock and 8x8 block: |mvd| x, y (capped, x264 keeps 8 bits); list 0 / list 1
    int last_dqp = 0, prev_coded_qp;    // mb_qp_delta context: the previous macroblock's delta
    int nskip = 0;
    long mv_bits = 0, tex_bits = 0, pos_start = 0;          // i_mv_bits / i_tex_bits of the slice
    int lst = 0;                        // B slices: the list the motion helpers read
    int cur_direct = 0;                 // direct 8x8 blocks of the macroblock being coded

    CabacSlice(const SliceParams &sp, const x264gpu_mb *m, const int16_t *l, Cabac &c) : p(sp), mbs(m), levels(l), cb(c), amvd((size_t)sp.mbw * sp.mbh * 8, 0),
        amvd1(sp.slice_type == X264GPU_SLICE_B ? (size_t)sp.mbw * sp.mbh * 8 : 0, 0), prev_coded_qp(sp.qp) {}

    const x264gpu_mb *left(int mbx, int mby) const { return mbx > 0 ? &mbs[mby * p.mbw + mbx - 1] : nullptr; }
    const x264gpu_mb *top(int mbx, int mby) const { return mby > p.first_row ? &mbs[(mby - 1) * p.mbw + mbx] : nullptr; }      // not across the slice boundary

    // ---- motion vector prediction (8.4.1.3), 8x8 granular: identical in role to the CAVLC writer's ----
    struct Nb { bool avail; int ref; int mvx, mvy; };
    int cur_mb = 0, done8 = 0;
    Nb cur8[4];
    Nb block8(int gx, int gy) const
    {
        Nb n = { false, -1, 0, 0 };
        if (gx < 0 || gy < 2 * p.first_row || gx >= 2 * p.mbw || gy >= 2 * p.mbh) return n;
        const int i = (gy >> 1) * p.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i == cur_mb) { if (done8 >> k & 1) return cur8[k]; return n; }
        if (i > cur_mb) return n;
        n.avail = true;
        const x264gpu_mb &m = mbs[i];
        if (!is_intra(m)) {
            // a block that does not use the list: reference -1, zero vector (8.4.1.3.2: predFlagLX 0)
            if (lst) { n.ref = m.ref1[k]; if (n.ref >= 0) { n.mvx = m.mv1[k][0]; n.mvy = m.mv1[k][1]; } else n.ref = -1; }
            else { n.ref = m.ref[k]; if (n.ref >= 0) { n.mvx = m.mv[k][0]; n.mvy = m.mv[k][1]; } else n.ref = -1; }
        }
        return n;
    }
    void mvp_part(int mbx, int mby, int bx8, int by8, int w8, int shape, int part, int ref, int &px, int &py) const
    {
        const int gx = 2 * mbx + bx8, gy = 2 * mby + by8;
        Nb a = block8(gx - 1, gy), b = block8(gx, gy - 1), c = block8(gx + w8, gy - 1);
        if (!c.avail) c = block8(gx - 1, gy - 1);
        if (shape == 1) {
            if (part == 0 && b.ref == ref) { px = b.mvx; py = b.mvy; return; }
            if (part == 1 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
        } else if (shape == 2) {
            if (part == 0 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
            if (part == 1 && c.ref == ref) { px = c.mvx; py = c.mvy; return; }
        }
        if (!b.avail && !c.avail && a.avail) { b = a; c = a; }
        const int na = a.ref == ref, nb = b.ref == ref, nc = c.ref == ref;
        if (na + nb + nc == 1) { const Nb &s = na ? a : nb ? b : c; px = s.mvx; py = s.mvy; return; }
        auto med = [](int x, int y, int z) { int mn = x < y ? x : y, mx = x < y ? y : x; return z < mn ? mn : z > mx ? mx : z; };
        px = med(a.mvx, b.mvx, c.mvx); py = med(a.mvy, b.mvy, c.mvy);
    }
    // |mvd| of the 8x8 block at (gx, gy) for the context of mvd bins: 0 outside the slice, in intra / skipped macroblocks and for blocks of
    // the current macroblock not yet coded
    int amvd_at(int gx, int gy, int comp) const
    {
        if (gx < 0 || gy < 2 * p.first_row || gx >= 2 * p.mbw || gy >= 2 * p.mbh) return 0;
        const int i = (gy >> 1) * p.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i > cur_mb || (i == cur_mb && !(done8 >> k & 1))) return 0;
        return (lst ? amvd1 : amvd)[((size_t)i * 4 + k) * 2 + comp];
    }
    int ref_gt0_at(int gx, int gy) const
    {
        const Nb n = block8(gx, gy);
        if (!n.avail || n.ref <= 0) return 0;
        const int i = (gy >> 1) * p.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i == cur_mb) return !(cur_direct >> k & 1);
        if (is_b(mbs[i])) return b_use(mbs[i], k) != 3;      // B_Skip / B_Direct_16x16 / direct sub-macroblocks: refIdxZeroFlag's condTerm is 0
        return mbs[i].type != X264GPU_MB_P_SKIP;
    }

    int pred_i4_mode(int mbx, int mby, int blk) const
    {
        const int bx = kBlkX[blk], by = kBlkY[blk];
        int ma, mb;
        const x264gpu_mb &cur = mbs[mby * p.mbw + mbx];
        if (bx > 0) ma = cur.i4_mode[kIdxOf[by][bx - 1]];
        else if (mbx > 0) { const x264gpu_mb &n = mbs[mby * p.mbw + mbx - 1]; ma = (n.type == X264GPU_MB_I4x4 || n.type == X264GPU_MB_I8x8) ? n.i4_mode[kIdxOf[by][3]] : 2; }
        else return 2;
        if (by > 0) mb = cur.i4_mode[kIdxOf[by - 1][bx]];
        else if (mby > p.first_row) { const x264gpu_mb &n = mbs[(mby - 1) * p.mbw + mbx]; mb = (n.type == X264GPU_MB_I4x4 || n.type == X264GPU_MB_I8x8) ? n.i4_mode[kIdxOf[3][bx]] : 2; }
        else return 2;
        return ma < mb ? ma : mb;
    }

    // ---- coded_block_flag contexts (9.3.3.1.1.9) ----
    // does luma 4x4 block (bx, by) of macroblock m carry coefficients, as the neighbour rule sees it
    static int luma_cbf_of(const x264gpu_mb &m, int bx, int by)
    {
        if (is_skip(m)) return 0;
        const int b8 = (by >> 1) * 2 + (bx >> 1);
        if (!(m.cbp_luma >> b8 & 1)) return 0;
        if (m.transform8x8) return 1;                       // 8x8 transform: the flag of the 8x8 block is inferred 1
        return (m.nnz >> kIdxOf[by][bx]) & 1;
    }
    int cbf_inc_luma(int mbx, int mby, const x264gpu_mb &cur, int blk) const
    {
        const int bx = kBlkX[blk], by = kBlkY[blk], unavail = is_intra(cur) ? 1 : 0;
        int a, b;
        if (bx > 0) a = luma_cbf_of(cur, bx - 1, by); else { const x264gpu_mb *n = left(mbx, mby); a = n ? luma_cbf_of(*n, 3, by) : unavail; }
        if (by > 0) b = luma_cbf_of(cur, bx, by - 1); else { const x264gpu_mb *n = top(mbx, mby); b = n ? luma_cbf_of(*n, bx, 3) : unavail; }
        return a + 2 * b;
    }
    int cbf_inc_dc(int mbx, int mby, const x264gpu_mb &cur, int bit /* 24 luma DC, 25 U, 26 V */) const
    {
        const int unavail = is_intra(cur) ? 1 : 0;
        auto of = [&](const x264gpu_mb *n) {
            if (!n) return unavail;
            if (is_skip(*n)) return 0;
            if (bit == 24) return n->type == X264GPU_MB_I16x16 ? (int)((n->nnz >> 24) & 1) : 0;
            return n->cbp_chroma ? (int)((n->nnz >> bit) & 1) : 0;
        };
        return of(left(mbx, mby)) + 2 * of(top(mbx, mby));
    }
    int cbf_inc_chroma_ac(int mbx, int mby, const x264gpu_mb &cur, int c, int i) const
    {
        const int bx = i & 1, by = i >> 1, unavail = is_intra(cur) ? 1 : 0;
        auto of = [&](const x264gpu_mb &m, int x, int y) { return !is_skip(m) && m.cbp_chroma == 2 ? (int)((m.nnz >> (16 + c * 4 + y * 2 + x)) & 1) : 0; };
        int a, b;
        if (bx > 0) a = of(cur, 0, by); else { const x264gpu_mb *n = left(mbx, mby); a = n ? of(*n, 1, by) : unavail; }
        if (by > 0) b = of(cur, bx, 0); else { const x264gpu_mb *n = top(mbx, mby); b = n ? of(*n, bx, 1) : unavail; }
        return a + 2 * b;
    }

    // ---- residual_block_cabac (7.3.5.3.3): l = levels in scan order, n = count (4, 15, 16 or 64) ----
    void residual(const int16_t *l, int cat)
    {
        switch (cat) { case 0: residual_t<0>(l); break; case 1: residual_t<1>(l); break; case 2: residual_t<2>(l); break; case 3: residual_t<3>(l); break; case 4: residual_t<4>(l); break; default: residual_t<5>(l); break; }
    }
    template <int cat> void residual_t(const int16_t *l)
    {
        static const int sig_off[6] = { 105, 120, 134, 149, 152, 402 }, last_off[6] = { 166, 181, 195, 210, 213, 417 }, abs_off[6] = { 227, 237, 247, 257, 266, 426 };
        static const int count_m1[6] = { 15, 14, 15, 3, 14, 63 };
        const int n1 = count_m1[cat];
        int last = n1;
        while (last > 0 && !l[last]) last--;
        int16_t coefs[64];
        int nc = 0;
        Cabac::R r = cb.regs();
        for (int i = 0;; i++) {
            const int so = cat == 5 ? cabac_sig8x8[i] : cat == 3 ? (i < 2 ? i : 2) : i, lo = cat == 5 ? cabac_last8x8[i] : cat == 3 ? (i < 2 ? i : 2) : i;
            if (l[i]) {
                coefs[nc++] = l[i];
                cb.decision(r, sig_off[cat] + so, 1);
                if (i == last) { cb.decision(r, last_off[cat] + lo, 1); break; }
                cb.decision(r, last_off[cat] + lo, 0);
            } else cb.decision(r, sig_off[cat] + so, 0);
            if (i + 1 == n1) { coefs[nc++] = l[n1]; break; }      // the last position needs no flags
        }
        // levels in reverse scan order; node = (coefficients equal to 1 seen, greater than 1 seen) folded into x264's node contexts
        static const uint8_t lvl1_ctx[8] = { 1, 2, 3, 4, 0, 0, 0, 0 }, gt1_ctx[8] = { 5, 5, 5, 5, 6, 7, 8, 9 };
        static const uint8_t trans[2][8] = { { 1, 2, 3, 3, 4, 5, 6, 7 }, { 4, 4, 4, 4, 5, 6, 7, 7 } };
        int node = 0;
        for (int k = nc - 1; k >= 0; k--) {
            const int v = coefs[k], a = v < 0 ? -v : v;
            const int ctx = abs_off[cat] + lvl1_ctx[node];
            if (a > 1) {
                cb.decision(r, ctx, 1);
                int g = gt1_ctx[node];
                if (cat == 3 && g > 8) g = 8;               // chroma DC has one context less
                const int c2 = abs_off[cat] + g;
                for (int i = (a < 15 ? a : 15) - 2; i > 0; i--) cb.decision(r, c2, 1);
                if (a < 15) cb.decision(r, c2, 0); else cb.ue_bypass(r, 0, a - 15);
                node = trans[1][node];
            } else { cb.decision(r, ctx, 0); node = trans[0][node]; }
            cb.bypass(r, v < 0);
        }
        cb.set_regs(r);
    }
    void block_cbf(const int16_t *l, int n, int cat, int inc)
    {
        static const int cbf_off[5] = { 85, 89, 93, 97, 101 };
        int nz = 0;
        for (int i = 0; i < n; i++) nz |= l[i];
        cb.decision(cbf_off[cat] + inc, nz != 0);
        if (nz) residual(l, cat);
    }

    void mb_type_intra(const x264gpu_mb &m, int c0, int c1, int c2, int c3, int c4, int c5)
    {
        if (m.type != X264GPU_MB_I16x16) { cb.decision(c0, 0); return; }
        cb.decision(c0, 1);
        cb.terminate(0);                                    // not I_PCM
        cb.decision(c1, m.cbp_luma != 0);
        if (!m.cbp_chroma) cb.decision(c2, 0);
        else { cb.decision(c2, 1); cb.decision(c3, m.cbp_chroma >> 1); }
        cb.decision(c4, m.i16_mode >> 1);
        cb.decision(c5, m.i16_mode & 1);
    }

    void mvd(int mbx, int mby, int b8, int w8, int h8, int comp, int val)
    {
        const int gx = 2 * mbx + (b8 & 1), gy = 2 * mby + (b8 >> 1);
        const int sum = amvd_at(gx - 1, gy, comp) + amvd_at(gx, gy - 1, comp);
        const int base = comp ? 47 : 40, inc = (sum > 2) + (sum > 32);
        const int a = val < 0 ? -val : val;
        if (!a) cb.decision(base + inc, 0);
        else {
            static const uint8_t ctxes[8] = { 3, 4, 5, 6, 6, 6, 6, 6 };
            cb.decision(base + inc, 1);
            if (a < 9) {
                for (int i = 1; i < a; i++) cb.decision(base + ctxes[i - 1], 1);
                cb.decision(base + ctxes[a - 1], 0);
            } else {
                for (int i = 1; i < 9; i++) cb.decision(base + ctxes[i - 1], 1);
                cb.ue_bypass(3, a - 9);
            }
            cb.bypass(val < 0);
        }
        const uint8_t capped = (uint8_t)(a < 66 ? a : 66);
        for (int y = b8 >> 1; y < (b8 >> 1) + h8; y++)
            for (int x = b8 & 1; x < (b8 & 1) + w8; x++) (lst ? amvd1 : amvd)[((size_t)cur_mb * 4 + y * 2 + x) * 2 + comp] = capped;
    }
    void ref_idx(int mbx, int mby, int b8, int ref)
    {
        const int gx = 2 * mbx + (b8 & 1), gy = 2 * mby + (b8 >> 1);
        int ctx = ref_gt0_at(gx - 1, gy) + 2 * ref_gt0_at(gx, gy - 1);
        for (int r = ref; r > 0; r--) { cb.decision(54 + ctx, 1); ctx = (ctx >> 2) + 4; }
        cb.decision(54 + ctx, 0);
    }

    // mb_type of a B slice (Table 9-37 b): the value's bin string; bin 0: ctxIdx 27 + ctx0, bin 1: 27 + 3, bin 2: 27 + 5 - b1, later bins: 27 + 5
    void mb_type_b(int value, int ctx0)
    {
        static const char *const bins[24] = { "0", "100", "101", "110000", "110001", "110010", "110011", "110100", "110101", "110110", "110111", "111110",
                                              "1110000", "1110001", "1110010", "1110011", "1110100", "1110101", "1110110", "1110111", "1111000", "1111001",
                                              "111111", "111101" /* the prefix of the intra types */ };
        const char *b = bins[value];
        for (int i = 0; b[i]; i++) cb.decision(i == 0 ? 27 + ctx0 : i == 1 ? 27 + 3 : i == 2 ? 27 + 5 - (b[1] - '0') : 27 + 5, b[i] - '0');
    }
    void sub_mb_type_b(int use)          // B_Direct_8x8 "0", B_L0_8x8 "100", B_L1_8x8 "101", B_Bi_8x8 "11000"; ctxIdx 36, 37, 38 | 39 (by b1), 39
    {
        if (use == 3) { cb.decision(36, 0); return; }
        cb.decision(36, 1);
        if (use == 2) { cb.decision(37, 1); cb.decision(38, 0); cb.decision(39, 0); cb.decision(39, 0); return; }
        cb.decision(37, 0);
        cb.decision(39, use == 1);
    }
    // mb_type, sub_mb_type, ref_idx_l0, ref_idx_l1, mvd_l0, mvd_l1 of an inter macroblock of a B slice (7.3.5.1 / 7.3.5.2)
    void mb_pred_b(int mbx, int mby, const x264gpu_mb &m, int ctx0)
    {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
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
            // Table 7-14: 4 + 2 * pair + (8x16); pairs ordered L0_L0, L1_L1, L0_L1, L1_L0, L0_Bi, L1_Bi, Bi_L0, Bi_L1, Bi_Bi
            static const int8_t pair_of[3][3] = { { 0, 2, 4 }, { 3, 1, 5 }, { 6, 7, 8 } };
            mb_type_b(4 + 2 * pair_of[use[0]][use[1]] + (part == 2), ctx0);
        }
        for (lst = 0; lst < 2; lst++) {
            if ((lst ? p.num_ref1 : p.num_ref) <= 1) continue;
            done8 = 0;
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[part][k];
                const int b8 = g[1] * 2 + g[0];
                const bool sends = !(use[k] == 3 || use[k] == 1 - lst);
                const int r = lst ? m.ref1[b8] : m.ref[b8];
                if (sends) ref_idx(mbx, mby, b8, r);
                for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur8[yy * 2 + xx] = Nb{ true, sends || use[k] == 3 ? r : -1, 0, 0 }; done8 |= 1 << (yy * 2 + xx); }
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
                    mvp_part(mbx, mby, g[0], g[1], g[2], part, k, r, px, py);
                    mvd(mbx, mby, b8, g[2], g[3], 0, vx - px);
                    mvd(mbx, mby, b8, g[2], g[3], 1, vy - py);
                }
                for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur8[yy * 2 + xx] = Nb{ true, r >= 0 ? r : -1, r >= 0 ? vx : 0, r >= 0 ? vy : 0 }; done8 |= 1 << (yy * 2 + xx); }
            }
        }
        lst = 0;
    }

    void macroblock(int mbx, int mby)
    {
        const int i = mby * p.mbw + mbx;
        const x264gpu_mb &m = mbs[i];
        const int16_t *lv = mb_levels(levels, index, (size_t)i, lvbuf);
        const x264gpu_mb *L = left(mbx, mby), *T = top(mbx, mby);
        cur_mb = i; done8 = 0; lst = 0; cur_direct = 0;
        const bool bslice = p.slice_type == X264GPU_SLICE_B;
        const bool pslice = p.slice_type != X264GPU_SLICE_I && !bslice;
        if (bslice) {
            cb.decision(24 + (L && !is_skip(*L)) + (T && !is_skip(*T)), m.type == X264GPU_MB_B_SKIP);
            if (m.type == X264GPU_MB_B_SKIP) { nskip++; last_dqp = 0; return; }
        }
        if (pslice) {
            const int ctx = 11 + (L && L->type != X264GPU_MB_P_SKIP) + (T && T->type != X264GPU_MB_P_SKIP);
            cb.decision(ctx, m.type == X264GPU_MB_P_SKIP);
            if (m.type == X264GPU_MB_P_SKIP) { nskip++; last_dqp = 0; return; }
        }
        pos_start = cb.pos();
        const bool intra = is_intra(m);
        // ---- mb_type ----
        if (bslice) {
            const int ctx0 = (L && L->type != X264GPU_MB_B_SKIP && L->type != X264GPU_MB_B_DIRECT) + (T && T->type != X264GPU_MB_B_SKIP && T->type != X264GPU_MB_B_DIRECT);
            if (intra) { mb_type_b(23, ctx0); mb_type_intra(m, 32, 32 + 1, 32 + 2, 32 + 2, 32 + 3, 32 + 3); }
            else mb_pred_b(mbx, mby, m, ctx0);
        } else if (!pslice) {
            const int ctx = (L && L->type != X264GPU_MB_I4x4 && L->type != X264GPU_MB_I8x8) + (T && T->type != X264GPU_MB_I4x4 && T->type != X264GPU_MB_I8x8);
            mb_type_intra(m, 3 + ctx, 3 + 3, 3 + 4, 3 + 5, 3 + 6, 3 + 7);
        } else if (intra) {
            cb.decision(14, 1);
            mb_type_intra(m, 17, 17 + 1, 17 + 2, 17 + 2, 17 + 3, 17 + 3);
        } else if (m.partition == 3) { cb.decision(14, 0); cb.decision(15, 0); cb.decision(16, 1); }
        else {
            cb.decision(14, 0);
            if (m.partition == 0) { cb.decision(15, 0); cb.decision(16, 0); }
            else { cb.decision(15, 1); cb.decision(17, m.partition == 1); }
        }
        const int t8ctx = 399 + (L && L->transform8x8) + (T && T->transform8x8);
        if (intra) {
            if (m.type != X264GPU_MB_I16x16) {
                if (p.transform8x8_mode) cb.decision(t8ctx, m.type == X264GPU_MB_I8x8);
                for (int b = 0; b < 16; b += m.type == X264GPU_MB_I8x8 ? 4 : 1) {
                    const int pm = pred_i4_mode(mbx, mby, b);
                    int mode = m.i4_mode[b];
                    if (mode == pm) cb.decision(68, 1);
                    else {
                        cb.decision(68, 0);
                        if (mode > pm) mode--;
                        cb.decision(69, mode & 1); cb.decision(69, (mode >> 1) & 1); cb.decision(69, mode >> 2);
                    }
                }
            }
            // intra_chroma_pred_mode
            const int ctx = (L && is_intra(*L) && L->chroma_mode != 0) + (T && is_intra(*T) && T->chroma_mode != 0);
            if (!m.chroma_mode) cb.decision(64 + ctx, 0);
            else {
                cb.decision(64 + ctx, 1);
                cb.decision(64 + 3, m.chroma_mode > 1);
                if (m.chroma_mode > 1) cb.decision(64 + 3, m.chroma_mode > 2);
            }
        } else if (!bslice) {
            static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                                  { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
            const int nparts = m.partition == 0 ? 1 : m.partition == 3 ? 4 : 2;
            if (m.partition == 3) for (int k = 0; k < 4; k++) cb.decision(21, 1);            // sub_mb_type P_L0_8x8
            if (p.num_ref > 1)
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[m.partition][k];
                    const int b8 = g[1] * 2 + g[0];
                    ref_idx(mbx, mby, b8, m.ref[b8]);
                    // the reference of the partition's blocks becomes visible to the following partitions' ref_idx contexts
                    for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur8[yy * 2 + xx] = Nb{ true, m.ref[b8], 0, 0 }; done8 |= 1 << (yy * 2 + xx); }
                }
            done8 = 0;
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[m.partition][k];
                const int b8 = g[1] * 2 + g[0];
                int px, py;
                mvp_part(mbx, mby, g[0], g[1], g[2], m.partition, k, m.ref[b8], px, py);
                mvd(mbx, mby, b8, g[2], g[3], 0, m.mv[b8][0] - px);
                mvd(mbx, mby, b8, g[2], g[3], 1, m.mv[b8][1] - py);
                for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur8[yy * 2 + xx] = Nb{ true, m.ref[b8], m.mv[b8][0], m.mv[b8][1] }; done8 |= 1 << (yy * 2 + xx); }
            }
        }
        const long pos_tex = cb.pos();
        mv_bits += pos_tex - pos_start;
        struct TexCount { CabacSlice &s; long t0; ~TexCount() { s.tex_bits += s.cb.pos() - t0; } } tex_count{ *this, pos_tex };
        // ---- coded_block_pattern ----
        if (m.type != X264GPU_MB_I16x16) {
            for (int b8 = 0; b8 < 4; b8++) {
                const int x = b8 & 1, y = b8 >> 1;
                // condTerm = neighbouring 8x8 block has its cbp bit CLEAR (available, not I_PCM); skipped macroblocks have cbp 0
                const int a = x ? !((m.cbp_luma >> (b8 - 1)) & 1) : L ? !((L->cbp_luma >> (b8 + 1)) & 1) : 0;
                const int b = y ? !((m.cbp_luma >> (b8 - 2)) & 1) : T ? !((T->cbp_luma >> (b8 + 2)) & 1) : 0;
                cb.decision(73 + a + 2 * b, (m.cbp_luma >> b8) & 1);
            }
            const int ca = L && L->cbp_chroma, cbb = T && T->cbp_chroma;
            cb.decision(77 + ca + 2 * cbb, m.cbp_chroma != 0);
            if (m.cbp_chroma) cb.decision(77 + 4 + (L && L->cbp_chroma == 2) + 2 * (T && T->cbp_chroma == 2), m.cbp_chroma == 2);
        }
        if (!intra && p.transform8x8_mode && m.cbp_luma) cb.decision(t8ctx, m.transform8x8);
        if (m.cbp_luma || m.cbp_chroma || m.type == X264GPU_MB_I16x16) {
            // ---- mb_qp_delta ----
            int dqp = (int)m.qp - prev_coded_qp;
            int ctx = last_dqp != 0;
            if (dqp) {
                if (dqp < -26) dqp += 52; else if (dqp > 25) dqp -= 52;
                int val = dqp > 0 ? 2 * dqp - 1 : -2 * dqp;
                do { cb.decision(60 + ctx, 1); ctx = 2 + (ctx >> 1); } while (--val);
            }
            cb.decision(60 + ctx, 0);
            last_dqp = dqp;
            prev_coded_qp = m.qp;
            // ---- residual ----
            if (m.type == X264GPU_MB_I16x16) {
                block_cbf(lv + X264GPU_LV_LUMA_DC, 16, 0, cbf_inc_dc(mbx, mby, m, 24));
                if (m.cbp_luma) for (int b = 0; b < 16; b++) block_cbf(lv + b * 16 + 1, 15, 1, cbf_inc_luma(mbx, mby, m, b));
            } else if (m.transform8x8) {
                for (int i8 = 0; i8 < 4; i8++)
                    if ((m.cbp_luma >> i8) & 1) {
                        int16_t l8[64];
                        for (int z = 0; z < 64; z++) l8[z] = lv[(i8 * 4 + (z & 3)) * 16 + (z >> 2)];
                        residual(l8, 5);
                    }
            } else {
                for (int b = 0; b < 16; b++) if ((m.cbp_luma >> (b >> 2)) & 1) block_cbf(lv + b * 16, 16, 2, cbf_inc_luma(mbx, mby, m, b));
            }
            if (m.cbp_chroma) {
                for (int c = 0; c < 2; c++) block_cbf(lv + X264GPU_LV_CHROMA_DC + c * 4, 4, 3, cbf_inc_dc(mbx, mby, m, 25 + c));
                if (m.cbp_chroma == 2)
                    for (int c = 0; c < 2; c++)
                        for (int k = 0; k < 4; k++) block_cbf(lv + X264GPU_LV_CHROMA_AC + (c * 4 + k) * 16 + 1, 15, 4, cbf_inc_chroma_ac(mbx, mby, m, c, k));
            }
        } else last_dqp = 0;
    }
};

thread_local uint8_t g_last_states[460];     // diagnostics: the context variables the last slice written by this thread ended with

}  // namespace

// tests: (pStateIdx << 1) | valMPS of every context after the last CABAC slice this thread wrote — what a decoder holds at that point, and
// what the device's RD bit counter must have arrived at (x264 prices candidates on the states the finished macroblocks left behind)
extern "C" void x264host_cabac_last_states(uint8_t *out) { memcpy(out, g_last_states, sizeof(g_last_states)); }

void write_slice_cabac(std::vector<uint8_t> &out, const SliceParams &p, const x264gpu_mb *mbs, const int16_t *levels,
                       bool annexb, bool long_startcode, SliceStats *stats, const x264gpu_level_index *index)
{
    BitWriter bw;
    write_slice_header(bw, p);
    while (bw.bits() & 7) bw.put1(1);                  // cabac_alignment_one_bit
    Cabac cb;
    cb.head_bits = (long)bw.bits();
    cb.buf.reserve((size_t)p.mbw * (size_t)p.mbh * 24 + 64);
    cb.init(p.slice_type == X264GPU_SLICE_I, p.qp);
    CabacSlice s(p, mbs, levels, cb);
    s.index = index;
    const int i0 = p.first_row * p.mbw, i1 = (p.end_row > 0 ? p.end_row : p.mbh) * p.mbw;
    for (int i = i0; i < i1; i++) {
        s.macroblock(i % p.mbw, i / p.mbw);
        cb.terminate(i == i1 - 1);                          // end_of_slice_flag
    }
    if (stats) { stats->skip = s.nskip; stats->mv_bits = s.mv_bits; stats->tex_bits = s.tex_bits; }
    memcpy(g_last_states, cb.st, 460);
    std::vector<uint8_t> rbsp = bw.bytes();                  // (byte aligned; the flush wrote the stop bit and padded the last byte with zeros)
    rbsp.insert(rbsp.end(), cb.buf.begin(), cb.buf.end());
    append_nal(out, p.nal_ref_idc, p.idr ? 5 : 1, rbsp, annexb, long_startcode);
}

}  // namespace x264host

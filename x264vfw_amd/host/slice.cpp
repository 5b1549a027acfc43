// slice.cpp — host-side entropy coding of one slice from the GPU's macroblock records + levels
// (north star: "CABAC/entropy left on the host").  CAVLC (ITU-T H.264 7.3.4, 7.3.5, 9.2); plays the role
// of [x264-upstream] encoder/cavlc.c + the slice header of encoder/encoder.c behind x264_encoder_encode
// (reference call site codec.c:1693).  Motion-vector differences are formed here against the H.264 predictors
// (8.4.1.3) of the records' vectors; macroblock types, P_Skip included, are the device analysis' decisions.
#include "host.hpp"
#include "cavlc_tables.hpp"
#include <stdlib.h>
#include <thread>

namespace x264host {

namespace {

const uint8_t kBlkX[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
const uint8_t kBlkY[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
const uint8_t kIdxOf[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };  // [by][bx]

inline bool is_intra(const x264gpu_mb &m) { return m.type == X264GPU_MB_I4x4 || m.type == X264GPU_MB_I8x8 || m.type == X264GPU_MB_I16x16; }

struct SliceCtx {
    const SliceParams &p;
    const x264gpu_mb *mbs;
    const int16_t *levels;
    const x264gpu_level_index *index;          // the levels are packed (host.hpp mb_levels), or nullptr
    BitWriter &bw;
    uint8_t *tc;                  // total_coeff per block: [mb][24] (16 luma by block index, 4 U, 4 V); skip/absent = 0.  Filled for the
                                  // WHOLE picture before any row is coded (it depends on the macroblock's own levels only), so that
                                  // bands of rows can be coded by different threads
    int row0, row1;               // macroblock rows this context codes
    // skip-run bookkeeping across bands (P slices): macroblocks skipped before the band's first coded macroblock / after its last
    int lead_skip = 0, trail_skip = 0, nskip = 0;
    long mv_bits = 0, tex_bits = 0;          // i_mv_bits / i_tex_bits (the skip runs are 'misc')
    bool has_coded = false;

    int mbw() const { return p.mbw; }

    // mb_qp_delta (7.4.5): against QP_Y of the previous macroblock in decoding order — the records carry the settled quantisers
    // (a macroblock that sends no delta has inherited its predecessor's), the first macroblock predicts from the slice quantiser
    int qp_delta(const x264gpu_mb &m) const
    {
        const int i = (int)(&m - mbs);
        int d = (int)m.qp - (i == p.first_row * p.mbw ? p.qp : (int)mbs[i - 1].qp);
        // mb_qp_delta lies in [-26, +25] (7.4.5); QP_Y is recovered modulo 52, so a larger step wraps (x264 cavlc.c does the same)
        if (d < -26) d += 52; else if (d > 25) d -= 52;
        return d;
    }

    // ---- nC for coeff_token (9.2.1): average of left (A) and top (B) block totals ----
    int nc_luma(int mbx, int mby, int blk) const
    {
        int bx = kBlkX[blk], by = kBlkY[blk], na = -1, nb = -1;
        if (bx > 0) na = tc[(size_t)(mby * mbw() + mbx) * 24 + kIdxOf[by][bx - 1]];
        else if (mbx > 0) na = tc[(size_t)(mby * mbw() + mbx - 1) * 24 + kIdxOf[by][3]];
        if (by > 0) nb = tc[(size_t)(mby * mbw() + mbx) * 24 + kIdxOf[by - 1][bx]];
        else if (mby > p.first_row) nb = tc[(size_t)((mby - 1) * mbw() + mbx) * 24 + kIdxOf[3][bx]];
        if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
        return na >= 0 ? na : nb >= 0 ? nb : 0;
    }
    int nc_chroma(int mbx, int mby, int c, int i) const
    {
        int bx = i & 1, by = i >> 1, na = -1, nb = -1, base = 16 + c * 4;
        if (bx > 0) na = tc[(size_t)(mby * mbw() + mbx) * 24 + base + by * 2];
        else if (mbx > 0) na = tc[(size_t)(mby * mbw() + mbx - 1) * 24 + base + by * 2 + 1];
        if (by > 0) nb = tc[(size_t)(mby * mbw() + mbx) * 24 + base + bx];
        else if (mby > p.first_row) nb = tc[(size_t)((mby - 1) * mbw() + mbx) * 24 + base + 2 + bx];
        if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
        return na >= 0 ? na : nb >= 0 ? nb : 0;
    }

    // ---- residual_block_cavlc (7.3.5.3.2 / 9.2): l[0..n-1] in scan order; nC = -1 selects chroma DC ----
    int residual_block(const int16_t *l, int n, int nC)
    {
        int idx[16], total = 0;
        for (int i = 0; i < n; i++) if (l[i]) idx[total++] = i;      // ascending frequency
        int t1 = 0;
        for (int k = total - 1; k >= 0 && t1 < 3; k--) { if (abs(l[idx[k]]) == 1) t1++; else break; }
        if (nC < 0) bw.put(chroma_dc_coeff_token_bits[4 * total + t1], chroma_dc_coeff_token_len[4 * total + t1]);
        else {
            int tab = nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3;
            bw.put(coeff_token_bits[tab][4 * total + t1], coeff_token_len[tab][4 * total + t1]);
        }
        if (!total) return 0;
        for (int k = 0; k < t1; k++) bw.put1(l[idx[total - 1 - k]] < 0);
        int suffix_len = total > 10 && t1 < 3 ? 1 : 0;
        for (int k = total - 1 - t1; k >= 0; k--) {
            int level = l[idx[k]];
            int code = level > 0 ? 2 * level - 2 : -2 * level - 1;          // levelCode (9.2.2.1)
            if (k == total - 1 - t1 && t1 < 3) code -= 2;
            write_level(code, suffix_len);
            if (suffix_len == 0) suffix_len = 1;
            if (abs(level) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
        }
        if (total < n) {
            int zeros = idx[total - 1] + 1 - total;                         // total_zeros
            if (nC < 0) bw.put(chroma_dc_total_zeros_bits[total - 1][zeros], chroma_dc_total_zeros_len[total - 1][zeros]);
            else bw.put(total_zeros_bits[total - 1][zeros], total_zeros_len[total - 1][zeros]);
            int left = zeros;
            for (int k = total - 1; k > 0 && left > 0; k--) {
                int run = idx[k] - idx[k - 1] - 1;
                int t = (left < 7 ? left : 7) - 1;
                bw.put(run_before_bits[t][run], run_before_len[t][run]);
                left -= run;
            }
        }
        return total;
    }
    void write_level(int code, int suffix_len)
    {
        // level_prefix / level_suffix (9.2.2.1), including the escape forms
        if (suffix_len == 0) {
            if (code < 14) { bw.put(1, code + 1); return; }
            if (code < 30) { bw.put(1, 15); bw.put((uint32_t)(code - 14), 4); return; }
            code -= 30;                       // prefix 15: levelCode = 30 + suffix (12 bits)
        } else {
            if ((code >> suffix_len) < 15) { bw.put(1, (code >> suffix_len) + 1); bw.put((uint32_t)(code & ((1 << suffix_len) - 1)), suffix_len); return; }
            code -= 15 << suffix_len;         // prefix 15: 12-bit suffix
        }
        if (code < 4096) { bw.put(1, 16); bw.put((uint32_t)code, 12); return; }
        // prefix >= 16 (only reachable at very low QP): suffix size prefix-3, offset (1<<(prefix-3)) - 4096
        code -= 4096;
        int prefix = 16;
        while (code >= (1 << (prefix - 3))) { code -= 1 << (prefix - 3); prefix++; }
        bw.put(0, prefix); bw.put1(1);
        bw.put((uint32_t)code, prefix - 3);
    }

    // ---- motion vector prediction (8.4.1.3) on an 8x8-granular motion cache (smallest partition is 8x8) ----
    struct Nb { bool avail; int ref; int mvx, mvy; };
    int cur_mb = 0;               // macroblock being coded
    int done8 = 0;                // 8x8 blocks of the current macroblock whose motion is already known
    Nb cur8[4];                   // motion of the current macroblock's 8x8 blocks (valid where done8 is set)
    int lst = 0;                  // B slices: the list whose motion block8 / mvp_part look at
    Nb block8(int gx, int gy) const
    {
        Nb n = { false, -1, 0, 0 };
        if (gx < 0 || gy < 2 * p.first_row || gx >= 2 * p.mbw || gy >= 2 * p.mbh) return n;      // outside the picture or the slice
        int i = (gy >> 1) * p.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i == cur_mb) { if (done8 >> k & 1) return cur8[k]; return n; }
        if (i > cur_mb) return n;                          // raster order: everything before the current macroblock is available
        n.avail = true;
        const x264gpu_mb &m = mbs[i];
        if (!is_intra(m)) {
            if (lst) { n.ref = m.ref1[k]; n.mvx = m.ref1[k] >= 0 ? m.mv1[k][0] : 0; n.mvy = m.ref1[k] >= 0 ? m.mv1[k][1] : 0; }
            else { n.ref = m.ref[k]; n.mvx = m.ref[k] >= 0 ? m.mv[k][0] : 0; n.mvy = m.ref[k] >= 0 ? m.mv[k][1] : 0; }
        }
        return n;
    }
    // partition = 8x8 blocks [bx8, bx8+w8) x [by8, by8+h8) of macroblock (mbx,mby); shape/part select the
    // directional rules of 16x8 / 8x16 partitions
    void mvp_part(int mbx, int mby, int bx8, int by8, int w8, int shape, int part, int ref, int &px, int &py) const
    {
        int gx = 2 * mbx + bx8, gy = 2 * mby + by8;
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
        int na = a.ref == ref, nb = b.ref == ref, nc = c.ref == ref;
        if (na + nb + nc == 1) {
            const Nb &s = na ? a : nb ? b : c;
            px = s.mvx; py = s.mvy;
            return;
        }
        auto med = [](int x, int y, int z) { int mn = x < y ? x : y, mx = x < y ? y : x; return z < mn ? mn : z > mx ? mx : z; };
        px = med(a.mvx, b.mvx, c.mvx); py = med(a.mvy, b.mvy, c.mvy);
    }
    void pskip_mv(int mbx, int mby, int &px, int &py) const
    {
        Nb a = block8(2 * mbx - 1, 2 * mby), b = block8(2 * mbx, 2 * mby - 1);
        if (!a.avail || !b.avail || (a.ref == 0 && a.mvx == 0 && a.mvy == 0) || (b.ref == 0 && b.mvx == 0 && b.mvy == 0)) { px = py = 0; return; }
        mvp_part(mbx, mby, 0, 0, 2, 0, 0, 0, px, py);
    }

    // predicted intra 4x4 mode (8.3.1.1)
    int pred_i4_mode(int mbx, int mby, int blk) const
    {
        int bx = kBlkX[blk], by = kBlkY[blk], ma, mb;
        const x264gpu_mb &cur = mbs[mby * p.mbw + mbx];
        if (bx > 0) ma = cur.i4_mode[kIdxOf[by][bx - 1]];
        else if (mbx > 0) { const x264gpu_mb &n = mbs[mby * p.mbw + mbx - 1]; ma = (n.type == X264GPU_MB_I4x4 || n.type == X264GPU_MB_I8x8) ? n.i4_mode[kIdxOf[by][3]] : 2; }
        else return 2;
        if (by > 0) mb = cur.i4_mode[kIdxOf[by - 1][bx]];
        else if (mby > p.first_row) { const x264gpu_mb &n = mbs[(mby - 1) * p.mbw + mbx]; mb = (n.type == X264GPU_MB_I4x4 || n.type == X264GPU_MB_I8x8) ? n.i4_mode[kIdxOf[3][bx]] : 2; }
        else return 2;
        return ma < mb ? ma : mb;
    }

    void write_residual(int mbx, int mby, const x264gpu_mb &m, const int16_t *lv)
    {
        uint8_t *t = tc + (size_t)(mby * p.mbw + mbx) * 24;
        if (m.type == X264GPU_MB_I16x16) residual_block(lv + X264GPU_LV_LUMA_DC, 16, nc_luma(mbx, mby, 0));
        for (int i8 = 0; i8 < 4; i8++) {
            if (!(m.cbp_luma >> i8 & 1)) continue;
            for (int k = 0; k < 4; k++) {
                int b = i8 * 4 + k;
                if (m.type == X264GPU_MB_I16x16) t[b] = (uint8_t)residual_block(lv + b * 16 + 1, 15, nc_luma(mbx, mby, b));
                else t[b] = (uint8_t)residual_block(lv + b * 16, 16, nc_luma(mbx, mby, b));
            }
        }
        if (m.cbp_chroma) {
            for (int c = 0; c < 2; c++) residual_block(lv + X264GPU_LV_CHROMA_DC + c * 4, 4, -1);
            if (m.cbp_chroma == 2)
                for (int c = 0; c < 2; c++)
                    for (int i = 0; i < 4; i++)
                        t[16 + c * 4 + i] = (uint8_t)residual_block(lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16 + 1, 15, nc_chroma(mbx, mby, c, i));
        }
    }

    void write_mb_intra(int mbx, int mby, const x264gpu_mb &m, const int16_t *lv, int type_offset)
    {
        const long pos_start = (long)bw.bits();
        long pos_tex = pos_start;
        if (m.type == X264GPU_MB_I8x8) {
            // I_NxN with transform_size_8x8_flag = 1: four Intra8x8PredMode (8.3.2.1); modes are stored replicated over
            // the 8x8's 4x4 entries, so the 4x4 predictor of its top-left block is exactly predIntra8x8PredMode
            bw.ue(type_offset + 0);
            bw.put1(1);
            for (int i8 = 0; i8 < 4; i8++) {
                int pm = pred_i4_mode(mbx, mby, i8 * 4), mode = m.i4_mode[i8 * 4];
                if (mode == pm) bw.put1(1);
                else { bw.put1(0); bw.put((uint32_t)(mode < pm ? mode : mode - 1), 3); }
            }
            bw.ue(m.chroma_mode);
            pos_tex = (long)bw.bits();
            bw.ue(cbp_to_golomb_intra[m.cbp_luma | (m.cbp_chroma << 4)]);
            if (m.cbp_luma || m.cbp_chroma) bw.se(qp_delta(m));
        } else if (m.type == X264GPU_MB_I4x4) {
            bw.ue(type_offset + 0);
            if (p.transform8x8_mode) bw.put1(0);                 // transform_size_8x8_flag
            for (int b = 0; b < 16; b++) {
                int pm = pred_i4_mode(mbx, mby, b), mode = m.i4_mode[b];
                if (mode == pm) bw.put1(1);
                else { bw.put1(0); bw.put((uint32_t)(mode < pm ? mode : mode - 1), 3); }
            }
            bw.ue(m.chroma_mode);
            pos_tex = (long)bw.bits();
            bw.ue(cbp_to_golomb_intra[m.cbp_luma | (m.cbp_chroma << 4)]);
            if (m.cbp_luma || m.cbp_chroma) bw.se(qp_delta(m));  // mb_qp_delta
        } else {
            bw.ue(type_offset + 1 + m.i16_mode + 4 * m.cbp_chroma + (m.cbp_luma ? 12 : 0));
            bw.ue(m.chroma_mode);
            pos_tex = (long)bw.bits();
            bw.se(qp_delta(m));                                  // mb_qp_delta always present for Intra16x16
        }
        write_residual(mbx, mby, m, lv);
        mv_bits += pos_tex - pos_start; tex_bits += (long)bw.bits() - pos_tex;
    }

    // total_coeff of every block of one macroblock (what residual_block will return for the blocks write_residual codes)
    static void fill_tc(const x264gpu_mb &m, const int16_t *lv, uint8_t *t)
    {
        auto nzc = [](const int16_t *l, int n) { int c = 0; for (int i = 0; i < n; i++) c += l[i] != 0; return c; };
        for (int b = 0; b < 24; b++) t[b] = 0;
        const bool i16 = m.type == X264GPU_MB_I16x16;
        for (int i8 = 0; i8 < 4; i8++)
            if (m.cbp_luma >> i8 & 1)
                for (int k = 0; k < 4; k++) { const int b = i8 * 4 + k; t[b] = (uint8_t)(i16 ? nzc(lv + b * 16 + 1, 15) : nzc(lv + b * 16, 16)); }
        if (m.cbp_chroma == 2)
            for (int c = 0; c < 8; c++) t[16 + c] = (uint8_t)nzc(lv + X264GPU_LV_CHROMA_AC + c * 16 + 1, 15);
    }

    // an inter macroblock of a B slice (7.3.5 with Table 7-14 / 7-18): mb_type, sub_mb_type, ref_idx_l0 / _l1, mvd_l0 / _l1, then what every inter
    // macroblock carries.  Direct blocks send nothing; their inferred motion (in the record) is what the other partitions' predictors see
    void write_mb_b(int mbx, int mby, const x264gpu_mb &m, const int16_t *lv)
    {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        auto use_of = [&](int k) { return (m.type == X264GPU_MB_B_DIRECT || (m.type == X264GPU_MB_B_8x8 && (m.direct8 >> k & 1))) ? 3 : m.ref[k] >= 0 ? (m.ref1[k] >= 0 ? 2 : 0) : 1; };
        const long pos_start = (long)bw.bits();
        if (m.type == X264GPU_MB_B_DIRECT) bw.ue(0);
        else {
            const int part = m.partition & 3, nparts = part == 0 ? 1 : part == 3 ? 4 : 2;
            int use[4];
            for (int k = 0; k < nparts; k++) use[k] = use_of(geom[part][k][1] * 2 + geom[part][k][0]);
            if (part == 3) { bw.ue(22); for (int k = 0; k < 4; k++) bw.ue(use[k] == 3 ? 0 : (uint32_t)(use[k] + 1)); }
            else if (part == 0) bw.ue((uint32_t)(1 + use[0]));
            else {
                static const int8_t pair_of[3][3] = { { 0, 2, 4 }, { 3, 1, 5 }, { 6, 7, 8 } };
                bw.ue((uint32_t)(4 + 2 * pair_of[use[0]][use[1]] + (part == 2)));
            }
            for (int l = 0; l < 2; l++) {
                const int nact = l ? p.num_ref1 : p.num_ref;
                if (nact <= 1) continue;
                for (int k = 0; k < nparts; k++) {
                    const int b8 = geom[part][k][1] * 2 + geom[part][k][0];
                    if (use[k] == 2 || use[k] == l) bw.te(nact - 1, l ? m.ref1[b8] : m.ref[b8]);
                }
            }
            for (lst = 0; lst < 2; lst++) {
                done8 = 0;
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[part][k];
                    const int b8 = g[1] * 2 + g[0];
                    const int r = lst ? m.ref1[b8] : m.ref[b8], vx = lst ? m.mv1[b8][0] : m.mv[b8][0], vy = lst ? m.mv1[b8][1] : m.mv[b8][1];
                    if (use[k] == 2 || use[k] == lst) {
                        int px, py;
                        mvp_part(mbx, mby, g[0], g[1], g[2], part, k, r, px, py);
                        bw.se(vx - px); bw.se(vy - py);
                    }
                    for (int yy = g[1]; yy < g[1] + g[3]; yy++)
                        for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur8[yy * 2 + xx] = Nb{ true, r >= 0 ? r : -1, r >= 0 ? vx : 0, r >= 0 ? vy : 0 }; done8 |= 1 << (yy * 2 + xx); }
                }
            }
            lst = 0; done8 = 0;
        }
        const long pos_tex = (long)bw.bits();
        mv_bits += pos_tex - pos_start;
        bw.ue(cbp_to_golomb_inter[m.cbp_luma | (m.cbp_chroma << 4)]);
        if (p.transform8x8_mode && m.cbp_luma) bw.put1(m.transform8x8);      // (direct_8x8_inference: every block here is 8x8 or larger)
        if (m.cbp_luma || m.cbp_chroma) bw.se(qp_delta(m));
        write_residual(mbx, mby, m, lv);
        tex_bits += (long)bw.bits() - pos_tex;
    }

    // Codes rows [row0, row1).  In P slices the bits start at the first coded macroblock's mb_type: the mb_skip_run in front of it
    // (lead_skip + whatever the previous band left pending) is written by the caller that stitches the bands together.
    int16_t lvbuf[X264GPU_MB_LEVELS];
    void run()
    {
        int skip_run = 0;
        auto flush_run = [&] { if (!has_coded) { lead_skip = skip_run; has_coded = true; } else bw.ue(skip_run); skip_run = 0; };
        for (int mby = row0; mby < row1; mby++)
            for (int mbx = 0; mbx < p.mbw; mbx++) {
                int i = mby * p.mbw + mbx;
                const x264gpu_mb &m = mbs[i];
                const int16_t *lv = mb_levels(levels, index, (size_t)i, lvbuf);
                cur_mb = i; done8 = 0;
                if (p.slice_type == X264GPU_SLICE_I) write_mb_intra(mbx, mby, m, lv, 0);
                else if (is_intra(m)) { flush_run(); write_mb_intra(mbx, mby, m, lv, p.slice_type == X264GPU_SLICE_B ? 23 : 5); }
                else if (p.slice_type == X264GPU_SLICE_B) {
                    if (m.type == X264GPU_MB_B_SKIP) { skip_run++; nskip++; }
                    else { flush_run(); write_mb_b(mbx, mby, m, lv); }
                }
                else {
                    int px, py;
                    // P_Skip is the analysis' decision (x264_macroblock_analyse / the conversion at the end of x264_macroblock_encode); its record
                    // carries the skip vector, which the decoder re-derives (8.4.1.1) — the closed-loop tests compare the two
                    const bool skip = m.type == X264GPU_MB_P_SKIP;
                    if (skip) { skip_run++; nskip++; }
                    else {
                        // partition geometry in 8x8 units: {bx8, by8, w8, h8}
                        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } },
                                                              { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
                        const int nparts = m.partition == 0 ? 1 : m.partition == 3 ? 4 : 2;
                        flush_run();
                        const long pos_start = (long)bw.bits();
                        bw.ue(m.partition);                                  // P_L0_16x16 / P_L0_L0_16x8 / P_L0_L0_8x16 / P_8x8
                        if (m.partition == 3) for (int k = 0; k < 4; k++) bw.ue(0);      // sub_mb_type P_L0_8x8
                        if (p.num_ref > 1)
                            for (int k = 0; k < nparts; k++) { const int8_t *g = geom[m.partition][k]; bw.te(p.num_ref - 1, m.ref[g[1] * 2 + g[0]]); }
                        for (int k = 0; k < nparts; k++) {
                            const int8_t *g = geom[m.partition][k];
                            int b8 = g[1] * 2 + g[0];
                            mvp_part(mbx, mby, g[0], g[1], g[2], m.partition, k, m.ref[b8], px, py);
                            bw.se(m.mv[b8][0] - px); bw.se(m.mv[b8][1] - py);
                            for (int yy = g[1]; yy < g[1] + g[3]; yy++)
                                for (int xx = g[0]; xx < g[0] + g[2]; xx++) {
                                    cur8[yy * 2 + xx] = Nb{ true, m.ref[b8], m.mv[b8][0], m.mv[b8][1] };
                                    done8 |= 1 << (yy * 2 + xx);
                                }
                        }
                        const long pos_tex = (long)bw.bits();
                        mv_bits += pos_tex - pos_start;
                        bw.ue(cbp_to_golomb_inter[m.cbp_luma | (m.cbp_chroma << 4)]);
                        if (p.transform8x8_mode && m.cbp_luma) bw.put1(m.transform8x8);      // every partition here is >= 8x8
                        if (m.cbp_luma || m.cbp_chroma) bw.se(qp_delta(m));
                        write_residual(mbx, mby, m, lv);
                        tex_bits += (long)bw.bits() - pos_tex;
                    }
                }
            }
        if (has_coded) trail_skip = skip_run; else lead_skip = skip_run;
    }
};

}  // namespace

void write_slice_header(BitWriter &bw, const SliceParams &p)
{
    const bool islice = p.slice_type == X264GPU_SLICE_I, bslice = p.slice_type == X264GPU_SLICE_B;
    bw.ue((uint32_t)(p.first_row * p.mbw));                     // first_mb_in_slice
    bw.ue((islice ? 2 : bslice ? 1 : 0) + 5);                   // slice_type (+5: all slices of the picture alike)
    bw.ue(p.pps_id);
    bw.put((uint32_t)p.frame_num & ((1u << p.log2_max_frame_num) - 1), p.log2_max_frame_num);
    if (p.idr) bw.ue(p.idr_pic_id);
    if (p.log2_max_poc_lsb > 0) bw.put((uint32_t)p.poc & ((1u << p.log2_max_poc_lsb) - 1), p.log2_max_poc_lsb);      // pic_order_cnt_lsb (type 0)
    // pic_order_cnt_type 2: nothing to send
    if (bslice) bw.put1(p.direct_spatial);                      // direct_spatial_mv_pred_flag
    if (!islice) {
        const bool ovr = p.num_ref != p.num_ref_default || (bslice && p.num_ref1 != p.num_ref1_default);       // fewer pictures in the DPB than the PPS default
        bw.put1(ovr);                                           // num_ref_idx_active_override_flag
        if (ovr) { bw.ue(p.num_ref - 1); if (bslice) bw.ue(p.num_ref1 - 1); }
        for (int l = 0; l <= (bslice ? 1 : 0); l++) {
            bw.put1(p.reorder[l].n > 0);                        // ref_pic_list_modification_flag_lX
            if (p.reorder[l].n > 0) {
                for (int i = 0; i < p.reorder[l].n; i++) { bw.ue((uint32_t)p.reorder[l].cmd[i].idc); bw.ue((uint32_t)p.reorder[l].cmd[i].arg); }
                bw.ue(3);
            }
        }
    }
    if (p.weighted_pred && p.slice_type == X264GPU_SLICE_P) {   // pred_weight_table()
        int denom = 0;
        for (int i = p.num_ref - 1; i >= 0; i--) if (p.wl0[i].on) denom = p.wl0[i].denom;
        int cdenom = 0;
        for (int i = p.num_ref - 1; i >= 0; i--) if (p.wc0[i].on[0] || p.wc0[i].on[1]) cdenom = p.wc0[i].denom;
        bw.ue((uint32_t)denom);                                 // luma_log2_weight_denom
        bw.ue((uint32_t)cdenom);                                // chroma_log2_weight_denom
        for (int i = 0; i < p.num_ref; i++) {
            bw.put1(p.wl0[i].on != 0);                          // luma_weight_l0_flag
            if (p.wl0[i].on) { bw.se(p.wl0[i].scale); bw.se(p.wl0[i].offset); }
            const bool cw = p.wc0[i].on[0] || p.wc0[i].on[1];
            bw.put1(cw);                                        // chroma_weight_l0_flag
            if (cw) for (int c = 0; c < 2; c++) { bw.se(p.wc0[i].on[c] ? p.wc0[i].scale[c] : 1 << cdenom); bw.se(p.wc0[i].on[c] ? p.wc0[i].offset[c] : 0); }
        }
    }
    if (p.nal_ref_idc) {
        if (p.idr) { bw.put1(0); bw.put1(0); }                  // no_output_of_prior_pics, long_term_reference
        else {
            bw.put1(p.n_mmco > 0);                              // adaptive_ref_pic_marking_mode_flag
            if (p.n_mmco > 0) {
                for (int i = 0; i < p.n_mmco; i++) { bw.ue(1); bw.ue((uint32_t)(p.mmco_diff[i] - 1)); }      // operation 1: difference_of_pic_nums_minus1
                bw.ue(0);
            }
        }
    }
    if (p.cabac && !islice) bw.ue(0);                           // cabac_init_idc
    bw.se(p.qp - p.pic_init_qp);                                // slice_qp_delta
    bw.ue(p.disable_deblock_idc);
    if (p.disable_deblock_idc != 1) { bw.se(p.alpha_off_div2); bw.se(p.beta_off_div2); }
}

// Row bands are entropy-coded by `threads` threads (CAVLC has no state that crosses macroblocks except mb_skip_run, which the
// stitching below carries over; nC contexts come from the precomputed total_coeff table, predictors from the records), then
// their bit strings are concatenated behind the slice header.  The bytes do not depend on the number of threads.
void write_slice(std::vector<uint8_t> &out, const SliceParams &p_in, const x264gpu_mb *mbs, const int16_t *levels,
                 bool annexb, bool long_startcode, SliceStats *stats, int threads, const x264gpu_level_index *index)
{
    // x264_slice_write: "set the QP equal to the first QP in the slice for more accurate CABAC initialization" — the slice header carries the first
    // macroblock's quantiser (under AQ / macroblock-tree it differs from the picture's); the device started the slice's quantiser chain and its
    // context variables from the same value, and a first macroblock that codes nothing inherits exactly it
    SliceParams p = p_in;
    p.qp = mbs[(size_t)p.first_row * p.mbw].qp;
    if (p.cabac) { write_slice_cabac(out, p, mbs, levels, annexb, long_startcode, stats, index); return; }      // one arithmetic code word per slice: no row bands
    BitWriter bw;
    write_slice_header(bw, p);
    const size_t n = (size_t)p.mbw * p.mbh;
    const int r0 = p.first_row, r1 = p.end_row > 0 ? p.end_row : p.mbh, rows = r1 - r0;
    std::vector<uint8_t> tc(n * 24);
    int T = threads < 1 ? 1 : threads;
    if (T > rows / 4) T = rows / 4 > 0 ? rows / 4 : 1;              // at least four rows per band
    std::vector<BitWriter> bws((size_t)T);
    std::vector<SliceCtx> ctx;
    ctx.reserve((size_t)T);
    for (int t = 0; t < T; t++) {
        ctx.push_back(SliceCtx{ p, mbs, levels, index, bws[(size_t)t], tc.data(), r0 + (int)((long)rows * t / T), r0 + (int)((long)rows * (t + 1) / T) });
        bws[(size_t)t].reserve(n * 48 / (size_t)T + 64);
    }
    auto fill = [&](int t) {
        int16_t scratch[X264GPU_MB_LEVELS];
        for (size_t i = (size_t)ctx[(size_t)t].row0 * p.mbw; i < (size_t)ctx[(size_t)t].row1 * p.mbw; i++)
            SliceCtx::fill_tc(mbs[i], mb_levels(levels, index, i, scratch), tc.data() + i * 24);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(fill, t);          // pass 1: total_coeff of every block of the picture
    fill(0);
    for (auto &th : pool) th.join();
    pool.clear();
    for (int t = 1; t < T; t++) pool.emplace_back([&ctx, t] { ctx[(size_t)t].run(); });      // pass 2: the bands
    ctx[0].run();
    for (auto &th : pool) th.join();
    int carry = 0;                                                   // skipped macroblocks not yet covered by an mb_skip_run
    for (int t = 0; t < T; t++) {
        const SliceCtx &c = ctx[(size_t)t];
        if (p.slice_type == X264GPU_SLICE_I) { bw.append(bws[(size_t)t]); continue; }
        if (c.has_coded) { bw.ue((uint32_t)(carry + c.lead_skip)); bw.append(bws[(size_t)t]); carry = c.trail_skip; }
        else carry += c.lead_skip;
    }
    if (carry) bw.ue((uint32_t)carry);
    if (stats) { stats->skip = 0; stats->mv_bits = stats->tex_bits = 0; for (const SliceCtx &c : ctx) { stats->skip += c.nskip; stats->mv_bits += c.mv_bits; stats->tex_bits += c.tex_bits; } }
    bw.trailing();
    append_nal(out, p.nal_ref_idc, p.idr ? 5 : 1, bw.bytes(), annexb, long_startcode);
}

// first macroblock row of slice i of n (x264 slice threads split the rows evenly: threadslice start = (mb_height * i + n / 2) / n)
int slice_first_row(int mbh, int i, int n) { return (mbh * i + n / 2) / n; }

// One picture as `slices` slices, each its own NAL unit (offs gets the offset of each in `out`); slices share nothing, so they are coded by
// up to `threads` threads.  With more than one slice the loop filter stops at slice boundaries (disable_deblocking_filter_idc 2), as under
// x264's slice threads — unless p.slices_plain says they are --slices N slices, which x264 filters across (idc 0).
void write_picture(std::vector<uint8_t> &out, std::vector<size_t> *offs, const SliceParams &p, int slices, const x264gpu_mb *mbs, const int16_t *levels,
                   bool annexb, bool long_startcode_first, SliceStats *stats, int threads, const x264gpu_level_index *index)
{
    const int n = slices > 1 ? slices : 1;
    if (n == 1) { if (offs) offs->push_back(out.size()); write_slice(out, p, mbs, levels, annexb, long_startcode_first, stats, threads, index); return; }
    std::vector<std::vector<uint8_t>> parts((size_t)n);
    std::vector<SliceStats> st((size_t)n, SliceStats{ 0 });
    auto one = [&](int i) {
        SliceParams sp = p;
        sp.first_row = slice_first_row(p.mbh, i, n); sp.end_row = slice_first_row(p.mbh, i + 1, n);
        if (sp.disable_deblock_idc == 0 && !sp.slices_plain) sp.disable_deblock_idc = 2;
        write_slice(parts[(size_t)i], sp, mbs, levels, annexb, long_startcode_first && i == 0, &st[(size_t)i], 1, index);
    };
    int T = threads < 1 ? 1 : threads > n ? n : threads;
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back([&, t] { for (int i = t; i < n; i += T) one(i); });
    for (int i = 0; i < n; i += T) one(i);
    for (auto &th : pool) th.join();
    if (stats) { stats->skip = 0; stats->mv_bits = stats->tex_bits = 0; }
    for (int i = 0; i < n; i++) {
        if (offs) offs->push_back(out.size());
        out.insert(out.end(), parts[(size_t)i].begin(), parts[(size_t)i].end());
        if (stats) { stats->skip += st[(size_t)i].skip; stats->mv_bits += st[(size_t)i].mv_bits; stats->tex_bits += st[(size_t)i].tex_bits; }
    }
}

}  // namespace x264host

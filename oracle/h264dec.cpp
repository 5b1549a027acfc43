// oracle/h264dec.cpp — minimal H.264 decoder for the syntax subset the host entropy coder emits
// (TEST INFRASTRUCTURE ONLY; see x264o.h).  Closes the loop: bitstream -> pictures must equal the encoder's
// reconstruction bit for bit, which checks the slice/macroblock syntax (7.3.3-7.3.5), CAVLC (9.2), motion
// vector and intra-mode prediction (8.3.1.1, 8.4.1) and boundary strengths (8.7.2.1) independently of the
// encoder-side code paths.  Sample reconstruction reuses the oracle's spec-pinned DSP (predict, dequant,
// inverse transforms, interpolation, edge filters).  Subset: CAVLC or CABAC (cabac_dec.hpp; cabac_init_idc 0), frame MBs,
// I (I4x4 / I8x8 / I16x16) and P (16x16 / 16x8 / 8x16 / 8x8, P_Skip, intra) slices, one slice per picture, poc type 2, up to 4 refs.
#include "x264o.h"
#include "cavlc_dec.hpp"
#include "cabac_dec.hpp"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>



namespace {

struct BitReader {
    const uint8_t *p; size_t n; size_t pos = 0;          // pos in bits
    int get1() { if (pos >= n * 8) { err = true; return 0; } int b = (p[pos >> 3] >> (7 - (pos & 7))) & 1; pos++; return b; }
    uint32_t get(int k) { uint32_t v = 0; while (k--) v = (v << 1) | get1(); return v; }
    uint32_t ue() { int z = 0; while (!get1() && z < 32 && !err) z++; return z ? ((1u << z) - 1 + get(z)) : 0; }
    int se() { uint32_t k = ue(); return (k & 1) ? (int)((k + 1) >> 1) : -(int)(k >> 1); }
    bool more_rbsp_data() const
    {
        // true unless only the rbsp_trailing_bits remain
        if (pos >= n * 8) return false;
        size_t last = n;
        while (last > 0 && p[last - 1] == 0) last--;
        if (!last) return false;
        int b = 0;
        while (!((p[last - 1] >> b) & 1)) b++;
        size_t stop = (last - 1) * 8 + (7 - b);       // bit position of the stop bit
        return pos < stop;
    }
    bool err = false;
};

const uint8_t kBlkX[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
const uint8_t kBlkY[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
const uint8_t kIdxOf[4][4] = { { 0, 1, 4, 5 }, { 2, 3, 6, 7 }, { 8, 9, 12, 13 }, { 10, 11, 14, 15 } };

struct MbInfo {
    int intra, i16, skip;
    int t8;                  // transform_size_8x8_flag (luma residual / Intra_8x8)
    int mv8[4][2], ref8[4];  // motion per 8x8 block, list 0; ref -1 for intra / list not used
    int mv8b[4][2], ref8b[4]; // ... list 1 (B slices)
    int bmb;                 // macroblock of a B slice (both lists' fields are meaningful)
    int direct8;             // B: 8x8 blocks predicted by direct inference (B_Skip / B_Direct_16x16: 15)
    int cbp_is_direct16;     // B_Skip or B_Direct_16x16 (the neighbour term of mb_type's first bin)
    int qp;
    uint8_t i4mode[16];
    uint8_t tc[24];          // total_coeff per block (for nC)
    uint32_t nz;             // luma blocks with coefficients (deblock bS 2): bit per block idx
    // CABAC context derivation (9.3.3.1.1): what the neighbours' syntax elements were
    int cbp_luma, cbp_chroma, chroma_mode;
    uint32_t cbf;            // coded_block_flag per block: bits 0..15 luma 4x4 (block index), 16..23 chroma AC (plane * 4 + block), 24 luma DC, 25 / 26 chroma DC
    uint8_t amvd[4][2];      // |mvd| per 8x8 block and component, list 0
    uint8_t amvdb[4][2];     // ... list 1
    int slice;               // number of the slice the macroblock belongs to (availability 6.4.x, deblocking across slice edges)
};

struct Decoder {
    int mbw = 0, mbh = 0, width = 0, height = 0, crop_r = 0, crop_b = 0;
    int log2_max_frame_num = 4, poc_type = 2;
    int pic_init_qp = 26, chroma_qp_offset = 0, deblock_ctrl = 1, num_ref_default = 1, transform8x8_mode = 0, cabac = 0;
    int stride = 0, pad = 32, cpad = 16;
    size_t plane_bytes = 0, cplane_bytes = 0;
    std::vector<pixel> luma[8], chroma[8];   // picture slots: the references + the picture being decoded
    int cur = 0, slots = 2, have = 0, num_ref_frames = 1, nref_active = 1;
    // decoded picture buffer (8.2.4, 8.2.5): the short-term reference pictures in decoding order, the picture being decoded, the reference lists
    struct Ref { int slot, frame_num, poc; };
    std::vector<Ref> dpb;
    int log2_max_poc_lsb = 4, prev_poc_msb = 0, prev_poc_lsb = 0, cur_poc = 0, cur_frame_num = 0, cur_is_ref = 1;
    int weighted_bipred_idc = 0, num_ref1_default = 1, nref1_active = 0;
    // explicit weighted prediction of P slices (PPS weighted_pred_flag, pred_weight_table 7.3.3.2, samples 8.4.2.3.2)
    int weighted_pred = 0, luma_logwd = 0, chroma_logwd = 0;
    struct { int lw, lo, cw[2], co[2]; } wp[32];
    int list_slot[2][16], list_poc[2][16];
    std::vector<int> pending_mmco;           // picture numbers to mark unused once the picture is complete
    std::vector<MbInfo> slot_mb[8];          // motion of every kept picture (co-located blocks of direct prediction)
    int slot_lpoc[8][2][16];                 // ... and the POCs behind its reference lists when it was decoded (temporal direct: the picture a co-located block refers to)
    int direct_spatial = 1;                  // direct_spatial_mv_pred_flag of the slice
    std::vector<int> frame_pocs;             // POC of every output picture (decoding order)
    int weighted_slices[2] = { 0, 0 };       // diagnostics: P slices that carried an explicit luma weight / a chroma weight
    int next_mb = 0, slice_no = 0, pic_disable = 0, pic_a = 0, pic_b = 0;      // slices of the picture being decoded
    std::vector<int> mb_bits;        // CAVLC: bits of the macroblock layer of every macroblock, picture after picture (0 for skipped ones)
    int ref_slot(int r) const { return list_slot[0][r]; }
    int ref_slot_l(int l, int r) const { return list_slot[l][r]; }
    std::vector<MbInfo> mb;
    bool have_sps = false, have_pps = false;
    std::vector<std::vector<uint8_t>> frames;
    x264o_quant_tables qt;

    pixel *Y(int slot, int k = 0) { return luma[slot].data() + k * plane_bytes + (size_t)pad * stride + pad; }
    pixel *UV(int slot) { return chroma[slot].data() + (size_t)cpad * stride + 2 * cpad; }

    void alloc()
    {
        stride = (mbw * 16 + 2 * pad + 63) / 64 * 64;
        plane_bytes = (size_t)stride * (mbh * 16 + 2 * pad);
        cplane_bytes = (size_t)stride * (mbh * 8 + 2 * cpad);
        slots = num_ref_frames + 1;
        for (int s = 0; s < slots; s++) { luma[s].assign(4 * plane_bytes, 0); chroma[s].assign(cplane_bytes, 0); slot_mb[s].clear(); }
        dpb.clear();
        mb.assign((size_t)mbw * mbh, MbInfo());
        x264o_quant_init(&qt, 21, 11);
    }
};

int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

// ---- CAVLC parsing ----
int read_vlc(BitReader &br, const uint8_t *len, const uint16_t *bits, int n)
{
    // incremental prefix match (tables are prefix-free)
    uint32_t code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | br.get1();
        for (int i = 0; i < n; i++) if (len[i] == l && bits[i] == code) return i;
    }
    br.err = true;
    return 0;
}

// returns total_coeff; coefficients (scan order) written to out[0..max-1]
int residual_block(BitReader &br, int16_t *out, int maxn, int nC)
{
    memset(out, 0, sizeof(int16_t) * maxn);
    const cavlcdec::Tables &T = cavlcdec::tables();
    int tok = nC < 0 ? read_vlc(br, T.chroma_dc_coeff_token_len, T.chroma_dc_coeff_token_bits, 20)
                     : read_vlc(br, T.coeff_token_len[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3], T.coeff_token_bits[nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3], 68);
    int total = tok >> 2, t1 = tok & 3;
    if (!total) return 0;
    int level[16];
    int suffix_len = total > 10 && t1 < 3 ? 1 : 0;
    for (int i = 0; i < total; i++) {
        if (i < t1) { level[i] = br.get1() ? -1 : 1; continue; }
        int prefix = 0;
        while (!br.get1() && prefix < 32 && !br.err) prefix++;
        int size = prefix == 14 && suffix_len == 0 ? 4 : prefix >= 15 ? prefix - 3 : suffix_len;
        int code = ((prefix < 15 ? prefix : 15) << suffix_len) + (size ? (int)br.get(size) : 0);
        if (prefix >= 15 && suffix_len == 0) code += 15;
        if (prefix >= 16) code += (1 << (prefix - 3)) - 4096;
        if (i == t1 && t1 < 3) code += 2;
        level[i] = (code & 1) ? (-code - 1) >> 1 : (code + 2) >> 1;
        if (suffix_len == 0) suffix_len = 1;
        if (abs(level[i]) > (3 << (suffix_len - 1)) && suffix_len < 6) suffix_len++;
    }
    int zeros = 0;
    if (total < maxn) {
        if (nC < 0) zeros = read_vlc(br, T.chroma_dc_total_zeros_len[total - 1], T.chroma_dc_total_zeros_bits[total - 1], 4);
        else zeros = read_vlc(br, T.total_zeros_len[total - 1], T.total_zeros_bits[total - 1], 16);
    }
    int pos = total + zeros - 1;               // position of the highest-frequency coefficient
    int left = zeros;
    for (int i = 0; i < total; i++) {
        if (pos < 0 || pos >= maxn) { br.err = true; return total; }
        out[pos] = (int16_t)level[i];
        int run = 0;
        if (i < total - 1 && left > 0) { int t = (left < 7 ? left : 7) - 1; run = read_vlc(br, T.run_before_len[t], T.run_before_bits[t], 16); left -= run; }
        else if (i == total - 1) run = left;
        pos -= run + 1;
    }
    return total;
}

struct SliceDec {
    Decoder &d;
    BitReader &br;
    int slice_type, qp, disable_deblock, alpha_off, beta_off;
    int first_mb = 0, slice_no = 0, end_mb = 0;      // first macroblock of this slice, its number in the picture, one past its last macroblock (set by run)
    cabacdec::Engine cd;
    int last_dqp = 0;                // mb_qp_delta of the previous macroblock in decoding order (ctxIdxInc of the first bin)

    // ---- CABAC: binarisations + ctxIdxInc derivations (9.3.2, 9.3.3.1) ----
    const MbInfo *nbA(int mbx, int mby) const { return mbx > 0 ? &d.mb[mby * d.mbw + mbx - 1] : nullptr; }
    bool top_ok(int mbx, int mby) const { return mby > 0 && (mby - 1) * d.mbw + mbx >= first_mb; }      // the macroblock above belongs to this slice
    const MbInfo *nbB(int mbx, int mby) const { return top_ok(mbx, mby) ? &d.mb[(mby - 1) * d.mbw + mbx] : nullptr; }
    static bool is_nxn(const MbInfo &m) { return m.intra && !m.i16; }
    int ca_skip_flag(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        return cd.decision(11 + (a && !a->skip) + (b && !b->skip));
    }
    // mb_type of an intra macroblock (I-slice numbering: 0 = I_NxN, 1..24 = I_16x16 variants); c[] = the six context indices
    int ca_intra_type(int c0, int c1, int c2, int c3, int c4, int c5)
    {
        if (!cd.decision(c0)) return 0;
        if (cd.terminate()) { cd.err = true; return 0; }          // I_PCM: not in the subset
        const int luma = cd.decision(c1);
        int chroma = 0;
        if (cd.decision(c2)) chroma = cd.decision(c3) ? 2 : 1;
        const int hi = cd.decision(c4), lo = cd.decision(c5);
        return 1 + (hi * 2 + lo) + 4 * chroma + 12 * luma;
    }
    int ca_mb_type_i(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        return ca_intra_type(3 + (a && !is_nxn(*a)) + (b && !is_nxn(*b)), 6, 7, 8, 9, 10);
    }
    // P slices: 0..3 = P_L0_16x16, P_L0_L0_16x8, P_L0_L0_8x16, P_8x8; 5 + n = intra type n
    int ca_mb_type_p()
    {
        if (cd.decision(14)) return 5 + ca_intra_type(17, 18, 19, 19, 20, 20);
        if (!cd.decision(15)) return cd.decision(16) ? 3 : 0;
        return cd.decision(17) ? 1 : 2;
    }
    int ca_t8(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        return cd.decision(399 + (a && a->t8) + (b && b->t8));
    }
    int ca_i4_mode(int pm)
    {
        if (cd.decision(68)) return pm;
        int r = cd.decision(69); r |= cd.decision(69) << 1; r |= cd.decision(69) << 2;
        return r < pm ? r : r + 1;
    }
    int ca_chroma_mode(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        if (!cd.decision(64 + (a && a->intra && a->chroma_mode) + (b && b->intra && b->chroma_mode))) return 0;
        if (!cd.decision(67)) return 1;
        return cd.decision(67) ? 3 : 2;
    }
    int ca_cbp(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        int luma = 0;
        for (int b8 = 0; b8 < 4; b8++) {
            // condTermFlagN: the neighbouring 8x8 block is available and its coded_block_pattern bit is 0 (9.3.3.1.1.4)
            const int ca = (b8 & 1) ? !((luma >> (b8 - 1)) & 1) : a ? !((a->cbp_luma >> (b8 + 1)) & 1) : 0;
            const int cb_ = (b8 & 2) ? !((luma >> (b8 - 2)) & 1) : b ? !((b->cbp_luma >> (b8 + 2)) & 1) : 0;
            luma |= cd.decision(73 + ca + 2 * cb_) << b8;
        }
        int chroma = 0;
        if (cd.decision(77 + (a && a->cbp_chroma) + 2 * (b && b->cbp_chroma)))
            chroma = cd.decision(81 + (a && a->cbp_chroma == 2) + 2 * (b && b->cbp_chroma == 2)) ? 2 : 1;
        return luma | chroma << 4;
    }
    int ca_dqp()
    {
        int ctx = 60 + (last_dqp != 0), k = 0;
        while (cd.decision(ctx)) { ctx = k == 0 ? 62 : 63; if (++k > 104) { cd.err = true; break; } }
        const int v = (k & 1) ? (k + 1) >> 1 : -(k >> 1);
        last_dqp = v;
        return v;
    }
    // reference index / mvd contexts look at the 8x8 blocks left of and above the partition's first block
    int ref_gt0(int gx, int gy)
    {
        if (gx < 0 || gy < 0 || gx >= 2 * d.mbw || gy >= 2 * d.mbh) return 0;
        const int i = (gy >> 1) * d.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i > cur_idx || i < first_mb) return 0;                          // not decoded yet / another slice
        if (i == cur_idx) return (refs_known >> k & 1) && cur_refs[k] > 0;
        const MbInfo &m = d.mb[i];
        // predicted by direct inference (B_Skip, B_Direct_16x16, direct sub-macroblocks): refIdxZeroFlag's condTerm is 0
        return !m.intra && !m.skip && !(m.direct8 >> k & 1) && (lst ? m.ref8b[k] : m.ref8[k]) > 0;
    }
    int lst = 0;                     // B slices: the list the motion helpers read
    int ca_ref(int gx, int gy)
    {
        int ctx = ref_gt0(gx - 1, gy) + 2 * ref_gt0(gx, gy - 1), r = 0;
        while (cd.decision(54 + ctx)) { ctx = ctx < 4 ? 4 : 5; if (++r > 32) { cd.err = true; break; } }
        return r;
    }
    int amvd_of(int gx, int gy, int comp)
    {
        if (gx < 0 || gy < 0 || gx >= 2 * d.mbw || gy >= 2 * d.mbh) return 0;
        const int i = (gy >> 1) * d.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i < first_mb || i > cur_idx || (i == cur_idx && !(known8 >> k & 1))) return 0;
        return lst ? d.mb[i].amvdb[k][comp] : d.mb[i].amvd[k][comp];
    }
    int ca_mvd(int gx, int gy, int comp)
    {
        const int sum = amvd_of(gx - 1, gy, comp) + amvd_of(gx, gy - 1, comp), base = comp ? 47 : 40;
        if (!cd.decision(base + (sum > 2) + (sum > 32))) return 0;
        int a = 1;
        while (a < 9 && cd.decision(base + (a < 4 ? 2 + a : 6))) a++;
        if (a == 9) a += cd.golomb_bypass(3);
        return cd.bypass() ? -a : a;
    }
    // coded_block_flag neighbourhood (9.3.3.1.1.9): value of the flag of the neighbouring block, or the default for a missing one
    static int luma_flag(const MbInfo &m, int bx, int by)
    {
        if (m.skip || !((m.cbp_luma >> ((by >> 1) * 2 + (bx >> 1))) & 1)) return 0;
        return m.t8 ? 1 : (int)((m.cbf >> kIdxOf[by][bx]) & 1);
    }
    int cbf_ctx_luma(int mbx, int mby, const MbInfo &cur, int blk)
    {
        const int bx = kBlkX[blk], by = kBlkY[blk], missing = cur.intra ? 1 : 0;
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        const int fa = bx ? luma_flag(cur, bx - 1, by) : a ? luma_flag(*a, 3, by) : missing;
        const int fb = by ? luma_flag(cur, bx, by - 1) : b ? luma_flag(*b, bx, 3) : missing;
        return fa + 2 * fb;
    }
    int cbf_ctx_dc(int mbx, int mby, const MbInfo &cur, int bit)
    {
        const int missing = cur.intra ? 1 : 0;
        auto f = [&](const MbInfo *n) {
            if (!n) return missing;
            if (n->skip) return 0;
            if (bit == 24) return n->i16 ? (int)((n->cbf >> 24) & 1) : 0;
            return n->cbp_chroma ? (int)((n->cbf >> bit) & 1) : 0;
        };
        return f(nbA(mbx, mby)) + 2 * f(nbB(mbx, mby));
    }
    int cbf_ctx_chroma_ac(int mbx, int mby, const MbInfo &cur, int c, int i)
    {
        const int bx = i & 1, by = i >> 1, missing = cur.intra ? 1 : 0;
        auto f = [&](const MbInfo &m, int x, int y) { return !m.skip && m.cbp_chroma == 2 ? (int)((m.cbf >> (16 + c * 4 + y * 2 + x)) & 1) : 0; };
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        const int fa = bx ? f(cur, 0, by) : a ? f(*a, 1, by) : missing;
        const int fb = by ? f(cur, bx, 0) : b ? f(*b, bx, 1) : missing;
        return fa + 2 * fb;
    }
    // residual_block_cabac (7.3.5.3.3): cat = ctxBlockCat; out[0..n-1] in scan order; returns the number of non-zero levels
    int ca_coeffs(int16_t *out, int n, int cat)
    {
        static const int sig0[6] = { 105, 120, 134, 149, 152, 402 }, last0[6] = { 166, 181, 195, 210, 213, 417 }, abs0[6] = { 227, 237, 247, 257, 266, 426 };
        memset(out, 0, sizeof(int16_t) * (size_t)n);
        int pos[64], np = 0;
        for (int i = 0; i < n - 1; i++) {
            const int si = cat == 5 ? cabacdec::kSigInc8[i] : cat == 3 ? (i < 2 ? i : 2) : i, li = cat == 5 ? cabacdec::kLastInc8[i] : cat == 3 ? (i < 2 ? i : 2) : i;
            if (cd.decision(sig0[cat] + si)) {
                pos[np++] = i;
                if (cd.decision(last0[cat] + li)) goto levels;
            }
        }
        pos[np++] = n - 1;                                         // reached the last position: it is significant by inference
    levels:
        int eq1 = 0, gt1 = 0;
        for (int k = np - 1; k >= 0; k--) {
            const int c0 = abs0[cat] + (gt1 ? 0 : (eq1 + 1 < 4 ? eq1 + 1 : 4));
            int a = 1;
            if (cd.decision(c0)) {
                const int lim = cat == 3 ? 3 : 4, c1 = abs0[cat] + 5 + (gt1 < lim ? gt1 : lim);
                a = 2;
                while (a < 15 && cd.decision(c1)) a++;
                if (a == 15) a += cd.golomb_bypass(0);
                gt1++;
            } else eq1++;
            out[pos[k]] = (int16_t)(cd.bypass() ? -a : a);
        }
        return np;
    }
    // coded_block_flag + levels of one block; returns the number of non-zero levels (0 when the flag is 0)
    int ca_block(int16_t *out, int n, int cat, int inc)
    {
        static const int cbf0[5] = { 85, 89, 93, 97, 101 };
        memset(out, 0, sizeof(int16_t) * (size_t)n);
        if (!cd.decision(cbf0[cat] + inc)) return 0;
        return ca_coeffs(out, n, cat);
    }
    int cur_refs[4] = { 0, 0, 0, 0 }, refs_known = 0;

    int nc_luma(int mbx, int mby, int blk)
    {
        int bx = kBlkX[blk], by = kBlkY[blk], na = -1, nb = -1;
        if (bx > 0) na = d.mb[mby * d.mbw + mbx].tc[kIdxOf[by][bx - 1]];
        else if (mbx > 0) na = d.mb[mby * d.mbw + mbx - 1].tc[kIdxOf[by][3]];
        if (by > 0) nb = d.mb[mby * d.mbw + mbx].tc[kIdxOf[by - 1][bx]];
        else if (top_ok(mbx, mby)) nb = d.mb[(mby - 1) * d.mbw + mbx].tc[kIdxOf[3][bx]];
        return na >= 0 && nb >= 0 ? (na + nb + 1) >> 1 : na >= 0 ? na : nb >= 0 ? nb : 0;
    }
    int nc_chroma(int mbx, int mby, int c, int i)
    {
        int bx = i & 1, by = i >> 1, na = -1, nb = -1, base = 16 + c * 4;
        if (bx > 0) na = d.mb[mby * d.mbw + mbx].tc[base + by * 2];
        else if (mbx > 0) na = d.mb[mby * d.mbw + mbx - 1].tc[base + by * 2 + 1];
        if (by > 0) nb = d.mb[mby * d.mbw + mbx].tc[base + bx];
        else if (top_ok(mbx, mby)) nb = d.mb[(mby - 1) * d.mbw + mbx].tc[base + 2 + bx];
        return na >= 0 && nb >= 0 ? (na + nb + 1) >> 1 : na >= 0 ? na : nb >= 0 ? nb : 0;
    }

    struct Nb { bool avail; int ref, mvx, mvy; };
    int cur_idx = 0, known8 = 0;     // current macroblock; mask of its 8x8 blocks with decoded motion
    Nb blk8(int gx, int gy)
    {
        Nb n = { false, -1, 0, 0 };
        if (gx < 0 || gy < 0 || gx >= 2 * d.mbw || gy >= 2 * d.mbh) return n;
        int i = (gy >> 1) * d.mbw + (gx >> 1), k = (gy & 1) * 2 + (gx & 1);
        if (i < first_mb || i > cur_idx || (i == cur_idx && !(known8 >> k & 1))) return n;      // another slice, or not decoded yet
        n.avail = true;
        const MbInfo &m = d.mb[i];
        if (!m.intra) {
            if (lst) { n.ref = m.ref8b[k]; if (n.ref >= 0) { n.mvx = m.mv8b[k][0]; n.mvy = m.mv8b[k][1]; } else n.ref = -1; }
            else { n.ref = m.ref8[k]; if (n.ref >= 0) { n.mvx = m.mv8[k][0]; n.mvy = m.mv8[k][1]; } else n.ref = -1; }
        }
        return n;
    }
    // 8.4.1.3: gx,gy = first 8x8 block of the partition, w8 its width in 8x8 units; shape 1 = 16x8, 2 = 8x16
    void mvp(int gx, int gy, int w8, int shape, int part, int ref, int &px, int &py)
    {
        Nb a = blk8(gx - 1, gy), b = blk8(gx, gy - 1), c = blk8(gx + w8, gy - 1);
        if (!c.avail) c = blk8(gx - 1, gy - 1);
        if (shape == 1 && part == 0 && b.ref == ref) { px = b.mvx; py = b.mvy; return; }
        if (shape == 1 && part == 1 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
        if (shape == 2 && part == 0 && a.ref == ref) { px = a.mvx; py = a.mvy; return; }
        if (shape == 2 && part == 1 && c.ref == ref) { px = c.mvx; py = c.mvy; return; }
        if (!b.avail && !c.avail && a.avail) { b = a; c = a; }
        int cnt = (a.ref == ref) + (b.ref == ref) + (c.ref == ref);
        if (cnt == 1) { const Nb &s = a.ref == ref ? a : b.ref == ref ? b : c; px = s.mvx; py = s.mvy; return; }
        auto med = [](int x, int y, int z) { return x > y ? (y > z ? y : x > z ? z : x) : (x > z ? x : y > z ? z : y); };
        px = med(a.mvx, b.mvx, c.mvx); py = med(a.mvy, b.mvy, c.mvy);
    }
    void skip_mv(int mbx, int mby, int &px, int &py)
    {
        Nb a = blk8(2 * mbx - 1, 2 * mby), b = blk8(2 * mbx, 2 * mby - 1);
        if (!a.avail || !b.avail || (a.ref == 0 && !a.mvx && !a.mvy) || (b.ref == 0 && !b.mvx && !b.mvy)) { px = py = 0; return; }
        mvp(2 * mbx, 2 * mby, 2, 0, 0, 0, px, py);
    }

    // implicit bi-prediction weights (8.4.2.3.1): w0 for the list-0 sample, of 64
    int implicit_w0(int r0, int r1) const
    {
        if (d.weighted_bipred_idc != 2) return 32;
        const int poc0 = d.list_poc[0][r0], poc1 = d.list_poc[1][r1];
        const int tb = clampi(d.cur_poc - poc0, -128, 127), td = clampi(poc1 - poc0, -128, 127);
        if (td == 0) return 32;
        const int tx = (16384 + abs(td / 2)) / td, dsf = clampi((tb * tx + 32) >> 6, -1024, 1023);
        if ((dsf >> 2) < -64 || (dsf >> 2) > 128) return 32;
        return 64 - (dsf >> 2);
    }
    void pred_block(int mbx, int mby, int k, int l, int r, const int mv[2], pixel *y /* 8x8, stride 8 */, pixel *u, pixel *v /* 4x4, stride 4 */)
    {
        const int ref = d.ref_slot_l(l, r), ox = (k & 1) * 8, oy = (k >> 1) * 8;
        pixel *planes[4] = { d.Y(ref, 0), d.Y(ref, 1), d.Y(ref, 2), d.Y(ref, 3) };
        // 8.4.2.2: samples outside the picture are the nearest edge samples.  The planes carry a finite replicated border, so a vector that points
        // farther is pulled back to 24 samples outside the picture first: every sample the interpolation reads is an edge sample either way
        const int mvx = clampi(mv[0], 4 * (-16 * mbx - 24), 4 * (16 * (d.mbw - mbx - 1) + 24)), mvy = clampi(mv[1], 4 * (-16 * mby - 24), 4 * (16 * (d.mbh - mby - 1) + 24));
        x264o_mc_luma(y, 8, planes, d.stride, mbx * 16 + ox, mby * 16 + oy, mvx, mvy, 8, 8);
        x264o_mc_chroma(u, v, 4, d.UV(ref), d.stride, mbx * 8 + ox / 2, mby * 8 + oy / 2, mvx, mvy, 4, 4);
    }
    void inter_pred(int mbx, int mby, const MbInfo &m)
    {
        pixel pu[64], pv[64];
        pixel *rec = d.Y(d.cur) + (size_t)mby * 16 * d.stride + mbx * 16;
        for (int k = 0; k < 4; k++) {
            const int ox = (k & 1) * 8, oy = (k >> 1) * 8;
            pixel y0[64], u0[16], v0[16], y1[64], u1[16], v1[16];
            const bool use0 = m.ref8[k] >= 0, use1 = m.bmb && m.ref8b[k] >= 0;
            if (use0) pred_block(mbx, mby, k, 0, m.ref8[k], m.mv8[k], y0, u0, v0);
            if (use1) pred_block(mbx, mby, k, 1, m.ref8b[k], m.mv8b[k], y1, u1, v1);
            if (use0 && use1) {
                // weighted sample prediction, implicit mode (8.4.2.3): logWD 5, no offsets
                const int w0 = implicit_w0(m.ref8[k], m.ref8b[k]), w1 = 64 - w0;
                for (int i = 0; i < 64; i++) y0[i] = (pixel)clampi((y0[i] * w0 + y1[i] * w1 + 32) >> 6, 0, 255);
                for (int i = 0; i < 16; i++) { u0[i] = (pixel)clampi((u0[i] * w0 + u1[i] * w1 + 32) >> 6, 0, 255); v0[i] = (pixel)clampi((v0[i] * w0 + v1[i] * w1 + 32) >> 6, 0, 255); }
            }
            if (d.weighted_pred && !m.bmb && use0) {
                // explicit mode, one prediction (8.4.2.3.2): ((p * w + 2^(logWD - 1)) >> logWD) + o, or p * w + o when logWD is 0
                const auto &w = d.wp[m.ref8[k]];
                auto wt = [](pixel *p, int n, int lw, int lo, int logwd) {
                    for (int i = 0; i < n; i++) p[i] = (pixel)clampi(logwd >= 1 ? ((p[i] * lw + (1 << (logwd - 1))) >> logwd) + lo : p[i] * lw + lo, 0, 255);
                };
                wt(y0, 64, w.lw, w.lo, d.luma_logwd);
                wt(u0, 16, w.cw[0], w.co[0], d.chroma_logwd); wt(v0, 16, w.cw[1], w.co[1], d.chroma_logwd);
            }
            const pixel *sy = use0 ? y0 : y1, *su = use0 ? u0 : u1, *sv = use0 ? v0 : v1;
            for (int yy = 0; yy < 8; yy++) memcpy(rec + (size_t)(oy + yy) * d.stride + ox, sy + yy * 8, 8);
            for (int yy = 0; yy < 4; yy++) for (int xx = 0; xx < 4; xx++) { pu[(oy / 2 + yy) * 8 + ox / 2 + xx] = su[yy * 4 + xx]; pv[(oy / 2 + yy) * 8 + ox / 2 + xx] = sv[yy * 4 + xx]; }
        }
        pixel *uv = d.UV(d.cur) + (size_t)mby * 8 * d.stride + mbx * 16;
        for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) { uv[y * d.stride + 2 * x] = pu[y * 8 + x]; uv[y * d.stride + 2 * x + 1] = pv[y * 8 + x]; }
    }

    // ---- B slices (CABAC): 7.3.5 with Tables 7-14 / 7-18, binarisations of 9.3.2.5, spatial direct prediction of 8.4.1.2.2 ----
    int ca_skip_flag_b(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        return cd.decision(24 + (a && !a->skip) + (b && !b->skip));
    }
    // mb_type of a B slice: 0 B_Direct_16x16, 1..21 the predicted types of Table 7-14, 22 B_8x8, 23 + n = intra type n
    int ca_mb_type_b(int mbx, int mby)
    {
        const MbInfo *a = nbA(mbx, mby), *b = nbB(mbx, mby);
        auto nd = [](const MbInfo *n) { return n && !(n->bmb && !n->intra && n->direct8 == 15 && n->cbp_is_direct16); };
        if (!cd.decision(27 + nd(a) + nd(b))) return 0;
        if (!cd.decision(27 + 3)) return 1 + cd.decision(27 + 5);
        int v = cd.decision(27 + 4) << 3;
        v |= cd.decision(27 + 5) << 2; v |= cd.decision(27 + 5) << 1; v |= cd.decision(27 + 5);
        if (v < 8) return 3 + v;
        if (v == 13) return 23 + ca_intra_type(32, 33, 34, 34, 35, 35);
        if (v == 14) return 11;
        if (v == 15) return 22;
        const int b6 = cd.decision(27 + 5);
        return v == 12 ? 20 + b6 : 12 + (((v & 3) << 1) | b6);
    }
    int ca_sub_mb_type_b()           // 0 direct, 1 list 0, 2 list 1, 3 both; anything smaller than 8x8 is outside the subset
    {
        if (!cd.decision(36)) return 0;
        if (!cd.decision(37)) return 1 + cd.decision(39);
        if (cd.decision(38)) { cd.err = true; return 0; }
        if (cd.decision(39) || cd.decision(39)) { cd.err = true; return 0; }
        return 3;
    }
    // spatial direct prediction of the whole macroblock (8.4.1.2.2, direct_8x8_inference): fills ref8 / mv8 of both lists for the blocks in mask
    // temporal direct prediction (8.4.1.2.3, direct_8x8_inference: the corner block of each co-located 8x8): list 0 = the picture the co-located block
    // refers to (its list-0 motion, list 1's when list 0 is unused), found in this slice's list 0 by its POC; the co-located vector scaled by the
    // POC distances for list 0, the remainder for list 1 index 0; an intra co-located macroblock gives index 0 and zero vectors
    void direct_temporal(int mbx, int mby, MbInfo &m, int mask)
    {
        const int cs = d.ref_slot_l(1, 0);
        const std::vector<MbInfo> &colpic = d.slot_mb[cs];
        const MbInfo *col = colpic.empty() ? nullptr : &colpic[(size_t)(mby * d.mbw + mbx)];
        for (int k = 0; k < 4; k++) {
            if (!(mask >> k & 1)) continue;
            m.ref8b[k] = 0;
            if (!col || col->intra) { m.ref8[k] = 0; m.mv8[k][0] = m.mv8[k][1] = m.mv8b[k][0] = m.mv8b[k][1] = 0; continue; }
            const int lc = col->ref8[k] >= 0 ? 0 : 1;
            const int rc = lc ? col->ref8b[k] : col->ref8[k];
            const int *mc = lc ? col->mv8b[k] : col->mv8[k];
            if (rc < 0) { br.err = true; return; }
            const int poc_ref = d.slot_lpoc[cs][lc][rc];
            int r0 = -1;
            for (int j = 0; j < d.nref_active; j++) if (d.list_poc[0][j] == poc_ref) { r0 = j; break; }
            if (r0 < 0) { br.err = true; return; }          // (a conforming stream always finds it)
            const int poc0 = d.list_poc[0][r0], poc1 = d.list_poc[1][0];
            const int tb = clampi(d.cur_poc - poc0, -128, 127), td = clampi(poc1 - poc0, -128, 127);
            m.ref8[k] = r0;
            if (td == 0) { m.mv8[k][0] = mc[0]; m.mv8[k][1] = mc[1]; m.mv8b[k][0] = m.mv8b[k][1] = 0; continue; }
            const int tx = (16384 + abs(td / 2)) / td, dsf = clampi((tb * tx + 32) >> 6, -1024, 1023);
            m.mv8[k][0] = (dsf * mc[0] + 128) >> 8; m.mv8[k][1] = (dsf * mc[1] + 128) >> 8;
            m.mv8b[k][0] = m.mv8[k][0] - mc[0]; m.mv8b[k][1] = m.mv8[k][1] - mc[1];
        }
    }
    void direct_spatial(int mbx, int mby, MbInfo &m, int mask)
    {
        if (!d.direct_spatial) { direct_temporal(mbx, mby, m, mask); return; }
        int ref[2], mv[2][2];
        const int saved_known = known8;
        known8 = 0;                                      // the neighbours of the MACROBLOCK: nothing inside it counts
        for (int l = 0; l < 2; l++) {
            lst = l;
            const int gx = 2 * mbx, gy = 2 * mby;
            Nb a = blk8(gx - 1, gy), b = blk8(gx, gy - 1), c = blk8(gx + 2, gy - 1);
            if (!c.avail) c = blk8(gx - 1, gy - 1);
            auto minpos = [](int x, int y) { return x >= 0 && y >= 0 ? (x < y ? x : y) : (x > y ? x : y); };
            ref[l] = minpos(a.ref, minpos(b.ref, c.ref));
            mv[l][0] = mv[l][1] = 0;
            if (ref[l] >= 0) mvp(gx, gy, 2, 0, 0, ref[l], mv[l][0], mv[l][1]);
        }
        lst = 0;
        known8 = saved_known;
        const bool zero_pred = ref[0] < 0 && ref[1] < 0;
        if (zero_pred) ref[0] = ref[1] = 0;
        // the co-located macroblock in the first picture of list 1
        const std::vector<MbInfo> &colpic = d.slot_mb[d.ref_slot_l(1, 0)];
        const MbInfo *col = colpic.empty() ? nullptr : &colpic[(size_t)(mby * d.mbw + mbx)];
        for (int k = 0; k < 4; k++) {
            if (!(mask >> k & 1)) continue;
            bool col_zero = false;
            if (col && !col->intra) {
                // the corner block of the co-located 8x8: its list-0 motion, or list 1's when list 0 is unused
                const int rc = col->ref8[k] >= 0 ? col->ref8[k] : col->bmb ? col->ref8b[k] : -1;
                const int *mc = col->ref8[k] >= 0 ? col->mv8[k] : col->mv8b[k];
                col_zero = rc == 0 && mc[0] >= -1 && mc[0] <= 1 && mc[1] >= -1 && mc[1] <= 1;
            }
            for (int l = 0; l < 2; l++) {
                int *dref = l ? m.ref8b : m.ref8; int (*dmv)[2] = l ? m.mv8b : m.mv8;
                dref[k] = ref[l];
                if (ref[l] < 0 || zero_pred || (ref[l] == 0 && col_zero)) { dmv[k][0] = dmv[k][1] = 0; }
                else { dmv[k][0] = mv[l][0]; dmv[k][1] = mv[l][1]; }
            }
        }
    }
    void skipped_mb_b(int i)
    {
        MbInfo &m = d.mb[i];
        const int mbx = i % d.mbw, mby = i / d.mbw;
        cur_idx = i; known8 = 0;
        m.intra = 0; m.skip = 1; m.bmb = 1; m.direct8 = 15; m.cbp_is_direct16 = 1; m.qp = qp; memset(m.i4mode, 2, 16);
        direct_spatial(mbx, mby, m, 15);
        inter_pred(mbx, mby, m);
    }
    // the residual of an inter macroblock, after its prediction: coded_block_pattern, transform_size_8x8_flag, mb_qp_delta, coefficients
    void inter_residual(int mbx, int mby, MbInfo &m)
    {
        int cbp = -1;
        if (d.cabac) cbp = ca_cbp(mbx, mby);
        else { int code = (int)br.ue(); if (code > 47) { br.err = true; return; } cbp = cavlcdec::kCbpOfCode[code][1]; }
        if (cbp < 0) { br.err = true; return; }
        m.cbp_luma = cbp & 15; m.cbp_chroma = cbp >> 4;
        m.t8 = (d.transform8x8_mode && (cbp & 15)) ? (d.cabac ? ca_t8(mbx, mby) : br.get1()) : 0;      // all partitions are >= 8x8 in this subset
        if (cbp) qp += d.cabac ? ca_dqp() : br.se();
        else last_dqp = 0;
        qp = (qp + 52) % 52;
        m.qp = qp;
        pixel *rec = d.Y(d.cur) + (size_t)mby * 16 * d.stride + mbx * 16;
        if (m.t8) {
            for (int i8 = 0; i8 < 4; i8++)
                if (cbp >> i8 & 1) luma8x8_residual(mbx, mby, i8, true, m, rec + (i8 >> 1) * 8 * d.stride + (i8 & 1) * 8);
        } else
        for (int b = 0; b < 16; b++) {
            if (!(cbp >> (b >> 2) & 1)) continue;
            int16_t l[16];
            if (d.cabac) { m.tc[b] = (uint8_t)ca_block(l, 16, 2, cbf_ctx_luma(mbx, mby, m, b)); if (m.tc[b]) m.cbf |= 1u << b; }
            else m.tc[b] = (uint8_t)residual_block(br, l, 16, nc_luma(mbx, mby, b));
            if (m.tc[b]) m.nz |= 1u << b;
            dctcoef blk[16];
            for (int k = 0; k < 16; k++) blk[x264o_zigzag4[k]] = l[k];
            x264o_dequant_4x4(blk, d.qt.dequant4_mf, qp);
            x264o_add4x4_idct(rec + kBlkY[b] * 4 * d.stride + kBlkX[b] * 4, d.stride, blk);
        }
        chroma_residual(mbx, mby, cbp >> 4, m, x264o_chroma_qp[clampi(qp + d.chroma_qp_offset, 0, 51)]);
    }

    void inter_mb_b(int mbx, int mby, int value, MbInfo &m)
    {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        // Table 7-14: prediction of the (at most two) partitions: 0 list 0, 1 list 1, 2 both
        static const int8_t pair[9][2] = { { 0, 0 }, { 1, 1 }, { 0, 1 }, { 1, 0 }, { 0, 2 }, { 1, 2 }, { 2, 0 }, { 2, 1 }, { 2, 2 } };
        m.intra = 0; m.i16 = 0; m.bmb = 1; m.direct8 = 0; m.cbp_is_direct16 = 0; memset(m.i4mode, 2, 16);
        m.cbf = 0; m.chroma_mode = 0;
        for (int k = 0; k < 4; k++) { m.ref8[k] = m.ref8b[k] = -1; m.mv8[k][0] = m.mv8[k][1] = m.mv8b[k][0] = m.mv8b[k][1] = 0; }
        int shape, use[4] = { 0, 0, 0, 0 };       // 0 list 0, 1 list 1, 2 both, 3 direct
        if (value == 0) { shape = 0; m.direct8 = 15; m.cbp_is_direct16 = 1; direct_spatial(mbx, mby, m, 15); known8 = 15; }
        else {
            if (value <= 3) { shape = 0; use[0] = value - 1; }
            else if (value <= 21) { shape = 1 + ((value - 4) & 1); use[0] = pair[(value - 4) >> 1][0]; use[1] = pair[(value - 4) >> 1][1]; }
            else {
                shape = 3;
                for (int k = 0; k < 4; k++) { const int t = d.cabac ? ca_sub_mb_type_b() : (int)br.ue(); if (t > 3) { br.err = true; return; } use[k] = t == 0 ? 3 : t - 1; if (t == 0) m.direct8 |= 1 << k; }
                if (m.direct8) direct_spatial(mbx, mby, m, m.direct8);
            }
            const int nparts = shape == 0 ? 1 : shape == 3 ? 4 : 2;
            int refs[2][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
            for (lst = 0; lst < 2; lst++) {
                refs_known = 0;
                const int nact = lst ? d.nref1_active : d.nref_active;
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[shape][k];
                    const bool sends = use[k] == 2 || use[k] == lst;
                    if (sends && nact > 1) { refs[lst][k] = d.cabac ? ca_ref(2 * mbx + g[0], 2 * mby + g[1]) : nact == 2 ? !br.get1() : (int)br.ue(); if (refs[lst][k] >= nact) { br.err = true; lst = 0; return; } }
                    for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur_refs[yy * 2 + xx] = sends ? refs[lst][k] : 0; refs_known |= 1 << (yy * 2 + xx); }
                }
            }
            for (lst = 0; lst < 2; lst++) {
                known8 = 0;
                for (int k = 0; k < nparts; k++) {
                    const int8_t *g = geom[shape][k];
                    const bool sends = use[k] == 2 || use[k] == lst;
                    int *dref = lst ? m.ref8b : m.ref8; int (*dmv)[2] = lst ? m.mv8b : m.mv8; uint8_t (*da)[2] = lst ? m.amvdb : m.amvd;
                    if (sends) {
                        int px, py;
                        // the partition's own blocks must not look like neighbours that use this reference yet: they are unknown until set below
                        mvp(2 * mbx + g[0], 2 * mby + g[1], g[2], shape, k, refs[lst][k], px, py);
                        const int dx = d.cabac ? ca_mvd(2 * mbx + g[0], 2 * mby + g[1], 0) : br.se(), dy = d.cabac ? ca_mvd(2 * mbx + g[0], 2 * mby + g[1], 1) : br.se();
                        for (int yy = g[1]; yy < g[1] + g[3]; yy++)
                            for (int xx = g[0]; xx < g[0] + g[2]; xx++) {
                                const int b8 = yy * 2 + xx;
                                dref[b8] = refs[lst][k]; dmv[b8][0] = px + dx; dmv[b8][1] = py + dy;
                                da[b8][0] = (uint8_t)(abs(dx) < 255 ? abs(dx) : 255); da[b8][1] = (uint8_t)(abs(dy) < 255 ? abs(dy) : 255);
                            }
                    }
                    for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) known8 |= 1 << (yy * 2 + xx);
                }
            }
            lst = 0;
            known8 = 15;
        }
        inter_pred(mbx, mby, m);
        inter_residual(mbx, mby, m);
    }

    void chroma_residual(int mbx, int mby, int cbp_chroma, MbInfo &m, int qpc)
    {
        int16_t dc[2][4] = { { 0 } }, ac[2][4][16];
        memset(ac, 0, sizeof(ac));
        if (d.cabac) {
            if (cbp_chroma) for (int c = 0; c < 2; c++) if (ca_block(dc[c], 4, 3, cbf_ctx_dc(mbx, mby, m, 25 + c))) m.cbf |= 1u << (25 + c);
            if (cbp_chroma == 2)
                for (int c = 0; c < 2; c++)
                    for (int i = 0; i < 4; i++) if (ca_block(ac[c][i] + 1, 15, 4, cbf_ctx_chroma_ac(mbx, mby, m, c, i))) m.cbf |= 1u << (16 + c * 4 + i);
        } else {
        if (cbp_chroma) for (int c = 0; c < 2; c++) residual_block(br, dc[c], 4, -1);
        if (cbp_chroma == 2)
            for (int c = 0; c < 2; c++)
                for (int i = 0; i < 4; i++) m.tc[16 + c * 4 + i] = (uint8_t)residual_block(br, ac[c][i] + 1, 15, nc_chroma(mbx, mby, c, i));
        }
        pixel *uv = d.UV(d.cur) + (size_t)mby * 8 * d.stride + mbx * 16;
        for (int c = 0; c < 2; c++) {
            dctcoef dq[4];
            x264o_dequant_2x2_dc(dq, dc[c], d.qt.dequant4_mf, qpc);
            pixel p[64];
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) p[y * 8 + x] = uv[y * d.stride + 2 * x + c];
            for (int i = 0; i < 4; i++) {
                dctcoef blk[16];
                for (int k = 0; k < 16; k++) blk[x264o_zigzag4[k]] = ac[c][i][k];
                x264o_dequant_4x4(blk, d.qt.dequant4_mf, qpc);
                blk[0] = dq[i];
                x264o_add4x4_idct(p + (i >> 1) * 32 + (i & 1) * 4, 8, blk);
            }
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) uv[y * d.stride + 2 * x + c] = p[y * 8 + x];
        }
    }

    // luma residual of one 8x8 block coded with the 8x8 transform: CAVLC carries it as four interleaved 4x4
    // blocks (7.3.5.3.2: lumaLevel8x8[64*i8 + 4*i + k] = level4x4[i8*4+k][i]); adds the residual to `r`
    void luma8x8_residual(int mbx, int mby, int i8, bool coded, MbInfo &m, pixel *r)
    {
        dctcoef c8[64];
        memset(c8, 0, sizeof(c8));
        bool any = false;
        if (coded && d.cabac) {
            int16_t l8[64];
            any = ca_coeffs(l8, 64, 5) != 0;                       // 4:2:0: no coded_block_flag for 8x8 blocks (inferred from the cbp bit)
            for (int z = 0; z < 64; z++) c8[x264o_zigzag8[z]] = l8[z];
        } else if (coded)
            for (int k = 0; k < 4; k++) {
                int16_t l[16];
                int b = i8 * 4 + k;
                m.tc[b] = (uint8_t)residual_block(br, l, 16, nc_luma(mbx, mby, b));
                if (m.tc[b]) any = true;
                for (int i = 0; i < 16; i++) c8[x264o_zigzag8[4 * i + k]] = l[i];
            }
        if (any) m.nz |= 0xfu << (4 * i8);           // 8.7.2.1: the 8x8 block containing the sample has coefficients
        x264o_dequant_8x8(c8, d.qt.dequant8_mf, qp);
        x264o_add8x8_idct8(r, d.stride, c8);
    }

    int i4_avail(int mbx, int mby, int b)
    {
        int bx = kBlkX[b], by = kBlkY[b], a = 0;
        const bool top = top_ok(mbx, mby), tl = mbx > 0 && top_ok(mbx - 1, mby), tr = mbx + 1 < d.mbw && top_ok(mbx + 1, mby);
        if (bx > 0 || mbx > 0) a |= X264O_AVAIL_LEFT;
        if (by > 0 || top) a |= X264O_AVAIL_TOP;
        if (bx > 0 ? (by > 0 || top) : by > 0 ? mbx > 0 : tl) a |= X264O_AVAIL_TOPLEFT;
        if (by == 0) { if (bx < 3 ? top : tr) a |= X264O_AVAIL_TOPRIGHT; }
        else if (bx < 3 && kIdxOf[by - 1][bx + 1] < b) a |= X264O_AVAIL_TOPRIGHT;
        return a;
    }
    int pred_i4(int mbx, int mby, int b, const MbInfo &cur)
    {
        int bx = kBlkX[b], by = kBlkY[b], ma, mb_;
        if (bx > 0) ma = cur.i4mode[kIdxOf[by][bx - 1]];
        else if (mbx > 0) { const MbInfo &n = d.mb[mby * d.mbw + mbx - 1]; ma = n.intra && !n.i16 ? n.i4mode[kIdxOf[by][3]] : 2; }
        else return 2;
        if (by > 0) mb_ = cur.i4mode[kIdxOf[by - 1][bx]];
        else if (top_ok(mbx, mby)) { const MbInfo &n = d.mb[(mby - 1) * d.mbw + mbx]; mb_ = n.intra && !n.i16 ? n.i4mode[kIdxOf[3][bx]] : 2; }
        else return 2;
        return ma < mb_ ? ma : mb_;
    }

    void intra_mb(int mbx, int mby, int mbtype /* I-slice numbering */, MbInfo &m)
    {
        pixel *rec = d.Y(d.cur) + (size_t)mby * 16 * d.stride + mbx * 16;
        int left = mbx > 0, top = top_ok(mbx, mby);
        m.intra = 1; for (int k = 0; k < 4; k++) { m.ref8[k] = -1; m.mv8[k][0] = m.mv8[k][1] = 0; }
        int cbp_luma = 0, cbp_chroma = 0, i16mode = 0;
        if (mbtype == 0) {
            m.i16 = 0;
            m.t8 = d.transform8x8_mode ? (d.cabac ? ca_t8(mbx, mby) : br.get1()) : 0;
            if (m.t8) {
                // Intra8x8PredMode (8.3.2.1): predicted from the neighbouring 4x4 entries; stored replicated
                for (int i8 = 0; i8 < 4; i8++) {
                    int pm = pred_i4(mbx, mby, i8 * 4, m), mode;
                    if (d.cabac) mode = ca_i4_mode(pm);
                    else if (br.get1()) mode = pm;
                    else { int r = (int)br.get(3); mode = r < pm ? r : r + 1; }
                    memset(m.i4mode + i8 * 4, mode, 4);
                }
            } else
            for (int b = 0; b < 16; b++) {
                int pm = pred_i4(mbx, mby, b, m);
                if (d.cabac) m.i4mode[b] = (uint8_t)ca_i4_mode(pm);
                else if (br.get1()) m.i4mode[b] = (uint8_t)pm;
                else { int r = (int)br.get(3); m.i4mode[b] = (uint8_t)(r < pm ? r : r + 1); }
            }
        } else {
            m.i16 = 1;
            int t = mbtype - 1;
            i16mode = t & 3; cbp_chroma = (t >> 2) % 3; cbp_luma = t >= 12 ? 15 : 0;
            memset(m.i4mode, 2, 16);
        }
        int chroma_mode = d.cabac ? ca_chroma_mode(mbx, mby) : (int)br.ue();
        m.chroma_mode = chroma_mode;
        if (!m.i16) {
            int cbp = -1;
            if (d.cabac) cbp = ca_cbp(mbx, mby);
            else { int code = (int)br.ue(); if (code > 47) { br.err = true; return; } cbp = cavlcdec::kCbpOfCode[code][0]; }
            if (cbp < 0) { br.err = true; return; }
            cbp_luma = cbp & 15; cbp_chroma = cbp >> 4;
        }
        m.cbp_luma = cbp_luma; m.cbp_chroma = cbp_chroma;
        if (m.i16 || cbp_luma || cbp_chroma) qp += d.cabac ? ca_dqp() : br.se();
        else last_dqp = 0;
        qp = (qp + 52) % 52;
        m.qp = qp;
        int qpc = x264o_chroma_qp[clampi(qp + d.chroma_qp_offset, 0, 51)];
        if (m.i16) {
            int16_t dcl[16], acl[16][16];
            memset(acl, 0, sizeof(acl));
            if (d.cabac) {
                if (ca_block(dcl, 16, 0, cbf_ctx_dc(mbx, mby, m, 24))) m.cbf |= 1u << 24;
                for (int i8 = 0; i8 < 4; i8++)
                    if (cbp_luma >> i8 & 1)
                        for (int k = 0; k < 4; k++) { int b = i8 * 4 + k; if (ca_block(acl[b] + 1, 15, 1, cbf_ctx_luma(mbx, mby, m, b))) m.cbf |= 1u << b; }
            } else {
            residual_block(br, dcl, 16, nc_luma(mbx, mby, 0));
            for (int i8 = 0; i8 < 4; i8++)
                if (cbp_luma >> i8 & 1)
                    for (int k = 0; k < 4; k++) { int b = i8 * 4 + k; m.tc[b] = (uint8_t)residual_block(br, acl[b] + 1, 15, nc_luma(mbx, mby, b)); }
            }
            int mode = i16mode;
            if (mode == I_PRED_16x16_DC) mode = left && top ? I_PRED_16x16_DC : left ? I_PRED_16x16_DC_LEFT : top ? I_PRED_16x16_DC_TOP : I_PRED_16x16_DC_128;
            pixel pred[256];
            x264o_predict_16x16(pred, 16, rec, d.stride, mode);
            for (int y = 0; y < 16; y++) memcpy(rec + y * d.stride, pred + y * 16, 16);
            dctcoef dc[16];
            for (int k = 0; k < 16; k++) dc[x264o_zigzag4[k]] = dcl[k];
            x264o_idct4x4dc(dc);
            x264o_dequant_4x4_dc(dc, d.qt.dequant4_mf, qp);
            for (int b = 0; b < 16; b++) {
                dctcoef blk[16];
                for (int k = 0; k < 16; k++) blk[x264o_zigzag4[k]] = acl[b][k];
                x264o_dequant_4x4(blk, d.qt.dequant4_mf, qp);
                blk[0] = dc[kBlkY[b] * 4 + kBlkX[b]];
                x264o_add4x4_idct(rec + kBlkY[b] * 4 * d.stride + kBlkX[b] * 4, d.stride, blk);
            }
        } else if (m.t8) {
            for (int i8 = 0; i8 < 4; i8++) {
                int x8 = i8 & 1, y8 = i8 >> 1, avail = 0;
                if (x8 || left) avail |= X264O_AVAIL_LEFT;
                if (y8 || top) avail |= X264O_AVAIL_TOP;
                if ((x8 || left) && (y8 || top)) avail |= X264O_AVAIL_TOPLEFT;
                if (i8 == 0 ? top : i8 == 1 ? (top && mbx + 1 < d.mbw) : i8 == 2) avail |= X264O_AVAIL_TOPRIGHT;
                int mode = m.i4mode[i8 * 4];
                if (mode == I_PRED_4x4_DC) {
                    int l_ = avail & X264O_AVAIL_LEFT, t_ = avail & X264O_AVAIL_TOP;
                    mode = l_ && t_ ? I_PRED_4x4_DC : l_ ? I_PRED_4x4_DC_LEFT : t_ ? I_PRED_4x4_DC_TOP : I_PRED_4x4_DC_128;
                }
                pixel *r = rec + y8 * 8 * d.stride + x8 * 8, edge[33], p8[64];
                x264o_predict_8x8_filter(r, d.stride, edge, avail);
                x264o_predict_8x8(p8, 8, edge, mode);
                for (int y = 0; y < 8; y++) memcpy(r + y * d.stride, p8 + y * 8, 8);
                luma8x8_residual(mbx, mby, i8, cbp_luma >> i8 & 1, m, r);
            }
        } else {
            for (int b = 0; b < 16; b++) {
                int16_t l[16];
                memset(l, 0, sizeof(l));
                if (cbp_luma >> (b >> 2) & 1) {
                    if (d.cabac) { m.tc[b] = (uint8_t)ca_block(l, 16, 2, cbf_ctx_luma(mbx, mby, m, b)); if (m.tc[b]) m.cbf |= 1u << b; }
                    else m.tc[b] = (uint8_t)residual_block(br, l, 16, nc_luma(mbx, mby, b));
                }
                int avail = i4_avail(mbx, mby, b), mode = m.i4mode[b];
                if (mode == I_PRED_4x4_DC) {
                    int l_ = avail & X264O_AVAIL_LEFT, t_ = avail & X264O_AVAIL_TOP;
                    mode = l_ && t_ ? I_PRED_4x4_DC : l_ ? I_PRED_4x4_DC_LEFT : t_ ? I_PRED_4x4_DC_TOP : I_PRED_4x4_DC_128;
                }
                pixel *r = rec + kBlkY[b] * 4 * d.stride + kBlkX[b] * 4, p4[16];
                x264o_predict_4x4(p4, 4, r, d.stride, mode, avail);
                for (int y = 0; y < 4; y++) memcpy(r + y * d.stride, p4 + y * 4, 4);
                dctcoef blk[16];
                for (int k = 0; k < 16; k++) blk[x264o_zigzag4[k]] = l[k];
                x264o_dequant_4x4(blk, d.qt.dequant4_mf, qp);
                x264o_add4x4_idct(r, d.stride, blk);
                if (m.tc[b]) m.nz |= 1u << b;
            }
        }
        // chroma prediction
        pixel *uv = d.UV(d.cur) + (size_t)mby * 8 * d.stride + mbx * 16;
        int cmode = chroma_mode;
        if (cmode == I_PRED_CHROMA_DC) cmode = left && top ? I_PRED_CHROMA_DC : left ? I_PRED_CHROMA_DC_LEFT : top ? I_PRED_CHROMA_DC_TOP : I_PRED_CHROMA_DC_128;
        for (int c = 0; c < 2; c++) {
            pixel ring[81], out[64];
            memset(ring, 128, sizeof(ring));
            for (int y = -1; y < 8; y++)
                for (int x = -1; x < 8; x++) {
                    if ((y >= 0 && x >= 0) || (y < 0 && !top) || (x < 0 && !left)) continue;
                    ring[(y + 1) * 9 + x + 1] = uv[y * d.stride + 2 * x + c];
                }
            x264o_predict_8x8c(out, 8, ring + 10, 9, cmode);
            for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) uv[y * d.stride + 2 * x + c] = out[y * 8 + x];
        }
        chroma_residual(mbx, mby, cbp_chroma, m, qpc);
    }

    void inter_mb(int mbx, int mby, int shape, MbInfo &m)
    {
        static const int8_t geom[4][4][4] = { { { 0, 0, 2, 2 } }, { { 0, 0, 2, 1 }, { 0, 1, 2, 1 } }, { { 0, 0, 1, 2 }, { 1, 0, 1, 2 } },
                                              { { 0, 0, 1, 1 }, { 1, 0, 1, 1 }, { 0, 1, 1, 1 }, { 1, 1, 1, 1 } } };
        m.intra = 0; m.i16 = 0; memset(m.i4mode, 2, 16);
        int nparts = shape == 0 ? 1 : shape == 3 ? 4 : 2, refs[4] = { 0, 0, 0, 0 };
        m.cbf = 0; m.chroma_mode = 0;
        refs_known = 0;
        if (shape == 3) for (int k = 0; k < 4; k++) if (d.cabac ? !cd.decision(21) : br.ue() != 0) { br.err = true; return; }     // only P_L0_8x8 sub-macroblocks
        if (d.nref_active > 1)
            for (int k = 0; k < nparts; k++) {
                const int8_t *g = geom[shape][k];
                refs[k] = d.cabac ? ca_ref(2 * mbx + g[0], 2 * mby + g[1]) : d.nref_active == 2 ? !br.get1() : (int)br.ue();
                for (int yy = g[1]; yy < g[1] + g[3]; yy++) for (int xx = g[0]; xx < g[0] + g[2]; xx++) { cur_refs[yy * 2 + xx] = refs[k]; refs_known |= 1 << (yy * 2 + xx); }
            }
        for (int k = 0; k < nparts; k++) if (refs[k] >= d.nref_active) { br.err = true; return; }
        for (int k = 0; k < nparts; k++) {
            const int8_t *g = geom[shape][k];
            int px, py;
            mvp(2 * mbx + g[0], 2 * mby + g[1], g[2], shape, k, refs[k], px, py);
            const int dx = d.cabac ? ca_mvd(2 * mbx + g[0], 2 * mby + g[1], 0) : br.se(), dy = d.cabac ? ca_mvd(2 * mbx + g[0], 2 * mby + g[1], 1) : br.se();
            int mvx = px + dx, mvy = py + dy;
            if (getenv("X264O_DEC_DEBUG")) fprintf(stderr, "mb %d,%d shape %d part %d mvp %d,%d mv %d,%d\n", mbx, mby, shape, k, px, py, mvx, mvy);
            for (int yy = g[1]; yy < g[1] + g[3]; yy++)
                for (int xx = g[0]; xx < g[0] + g[2]; xx++) {
                    int b8 = yy * 2 + xx;
                    m.ref8[b8] = refs[k]; m.mv8[b8][0] = mvx; m.mv8[b8][1] = mvy;
                    m.amvd[b8][0] = (uint8_t)(abs(dx) < 255 ? abs(dx) : 255); m.amvd[b8][1] = (uint8_t)(abs(dy) < 255 ? abs(dy) : 255);
                    known8 |= 1 << b8;
                }
        }
        inter_pred(mbx, mby, m);
        inter_residual(mbx, mby, m);
    }

    void skipped_mb(int i)
    {
        MbInfo &m = d.mb[i];
        int mbx = i % d.mbw, mby = i / d.mbw, px, py;
        cur_idx = i; known8 = 0;
        skip_mv(mbx, mby, px, py);
        m.intra = 0; m.skip = 1; m.qp = qp; memset(m.i4mode, 2, 16);
        for (int k = 0; k < 4; k++) { m.ref8[k] = 0; m.mv8[k][0] = px; m.mv8[k][1] = py; }
        inter_pred(mbx, mby, m);
    }
    void run_cabac()
    {
        const int n = d.mbw * d.mbh;
        while (br.pos & 7) if (!br.get1()) { br.err = true; return; }          // cabac_alignment_one_bit
        cd.start(br.p, br.n, br.pos, slice_type != 2, qp);
        end_mb = -1;
        for (int i = first_mb; i < n && !cd.err && !br.err; i++) {
            const int mbx = i % d.mbw, mby = i / d.mbw;
            cur_idx = i; known8 = 0;
            d.mb[i] = MbInfo(); d.mb[i].slice = slice_no;
            if (slice_type == 0 && ca_skip_flag(mbx, mby)) { skipped_mb(i); last_dqp = 0; }
            else if (slice_type == 1 && ca_skip_flag_b(mbx, mby)) { if (getenv("X264O_DEC_DEBUG")) fprintf(stderr, "B mb %d,%d skip\n", mbx, mby); skipped_mb_b(i); last_dqp = 0; }
            else if (slice_type == 1) {
                MbInfo &m = d.mb[i];
                const int t = ca_mb_type_b(mbx, mby);
                if (getenv("X264O_DEC_DEBUG")) fprintf(stderr, "B mb %d,%d type %d\n", mbx, mby, t);
                if (t <= 22) inter_mb_b(mbx, mby, t, m);
                else intra_mb(mbx, mby, t - 23, m);
            } else {
                MbInfo &m = d.mb[i];
                const int t = slice_type == 0 ? ca_mb_type_p() : ca_mb_type_i(mbx, mby);
                if (slice_type == 0 && t <= 3) inter_mb(mbx, mby, t, m);
                else intra_mb(mbx, mby, slice_type == 0 ? t - 5 : t, m);
            }
            if (cd.terminate()) { end_mb = i + 1; break; }                       // end_of_slice_flag
        }
        if (cd.err || end_mb < 0) br.err = true;                                 // the picture ended without the flag
    }
    void run()
    {
        if (d.cabac) { run_cabac(); return; }
        int n = d.mbw * d.mbh, i = first_mb;
        while (i < n && !br.err) {
            if (slice_type == 1) {          // B slice: mb_skip_run of B_Skip macroblocks (direct prediction, nothing coded)
                int run = (int)br.ue();
                while (run-- && i < n) { d.mb[i] = MbInfo(); d.mb[i].slice = slice_no; skipped_mb_b(i); i++; }
                if (i >= n || !br.more_rbsp_data()) break;
            }
            if (slice_type == 0) {
                int run = (int)br.ue();
                while (run-- && i < n) {
                    MbInfo &m = d.mb[i];
                    m = MbInfo(); m.slice = slice_no;
                    int mbx = i % d.mbw, mby = i / d.mbw, px, py;
                    cur_idx = i; known8 = 0;
                    skip_mv(mbx, mby, px, py);
                    m.intra = 0; m.skip = 1; m.qp = qp; memset(m.i4mode, 2, 16);
                    for (int k = 0; k < 4; k++) { m.ref8[k] = 0; m.mv8[k][0] = px; m.mv8[k][1] = py; }
                    inter_pred(mbx, mby, m);
                    i++;
                }
                if (i >= n || !br.more_rbsp_data()) break;
            }
            MbInfo &m = d.mb[i];
            m = MbInfo(); m.slice = slice_no;
            int mbx = i % d.mbw, mby = i / d.mbw;
            const size_t pos0 = br.pos;
            int t = (int)br.ue();
            cur_idx = i; known8 = 0;
            if (slice_type == 0) { if (t <= 3) inter_mb(mbx, mby, t, m); else if (t >= 5) intra_mb(mbx, mby, t - 5, m); else br.err = true; }
            else if (slice_type == 1) { if (t <= 22) inter_mb_b(mbx, mby, t, m); else intra_mb(mbx, mby, t - 23, m); }
            else intra_mb(mbx, mby, t, m);
            d.mb_bits[d.mb_bits.size() - (size_t)(d.mbw * d.mbh) + (size_t)i] = (int)(br.pos - pos0);
            i++;
            if (!br.more_rbsp_data()) break;
        }
        end_mb = i;
    }

    // ---- deblocking (8.7), written from the clause; filters come from oracle/deblock.c ----
    int bs_of(const MbInfo &p, int pb, const MbInfo &q, int qb, bool mbedge)
    {
        if (p.intra || q.intra) return mbedge ? 4 : 3;
        if ((p.nz >> pb & 1) || (q.nz >> qb & 1)) return 2;
        int p8 = pb >> 2, q8 = qb >> 2;           // block index / 4 = 8x8 quadrant in H.264 block order
        if (!p.bmb && !q.bmb) {
            if (d.ref_slot_l(0, p.ref8[p8]) != d.ref_slot_l(0, q.ref8[q8])) return 1;       // different reference PICTURES (an index may repeat a picture)
            return abs(p.mv8[p8][0] - q.mv8[q8][0]) >= 4 || abs(p.mv8[p8][1] - q.mv8[q8][1]) >= 4;
        }
        // B slices (8.7.2.1): different reference PICTURES or a different number of vectors -> 1; else the vectors that point into the same picture
        // are compared (both pairings when the two pictures of a block coincide)
        int pp[2], qq[2]; const int *pm[2], *qm[2];
        int np = 0, nq = 0;
        if (p.ref8[p8] >= 0) { pp[np] = d.ref_slot_l(0, p.ref8[p8]); pm[np++] = p.mv8[p8]; }
        if (p.bmb && p.ref8b[p8] >= 0) { pp[np] = d.ref_slot_l(1, p.ref8b[p8]); pm[np++] = p.mv8b[p8]; }
        if (q.ref8[q8] >= 0) { qq[nq] = d.ref_slot_l(0, q.ref8[q8]); qm[nq++] = q.mv8[q8]; }
        if (q.bmb && q.ref8b[q8] >= 0) { qq[nq] = d.ref_slot_l(1, q.ref8b[q8]); qm[nq++] = q.mv8b[q8]; }
        auto far = [](const int *x, const int *y) { return abs(x[0] - y[0]) >= 4 || abs(x[1] - y[1]) >= 4; };
        if (np != nq) return 1;
        if (np == 1) return pp[0] != qq[0] || far(pm[0], qm[0]);
        if (!((pp[0] == qq[0] && pp[1] == qq[1]) || (pp[0] == qq[1] && pp[1] == qq[0]))) return 1;
        if (pp[0] != pp[1]) return pp[0] == qq[0] ? (far(pm[0], qm[0]) || far(pm[1], qm[1])) : (far(pm[0], qm[1]) || far(pm[1], qm[0]));
        return (far(pm[0], qm[0]) || far(pm[1], qm[1])) && (far(pm[0], qm[1]) || far(pm[1], qm[0]));
    }
    void deblock()
    {
        if (disable_deblock == 1) return;
        pixel *Yp = d.Y(d.cur), *UVp = d.UV(d.cur);
        for (int mby = 0; mby < d.mbh; mby++)
            for (int mbx = 0; mbx < d.mbw; mbx++) {
                const MbInfo &q = d.mb[mby * d.mbw + mbx];
                for (int vert = 1; vert >= 0; vert--)            // vertical edges first
                    for (int e = 0; e < 4; e++) {
                        const MbInfo *p = &q;
                        if ((e & 1) && q.t8) continue;           // transform_size_8x8_flag: no luma edge at 4-sample offsets
                        if (e == 0) {
                            if (vert) { if (!mbx) continue; p = &d.mb[mby * d.mbw + mbx - 1]; }
                            else { if (!mby) continue; p = &d.mb[(mby - 1) * d.mbw + mbx]; }
                            if (disable_deblock == 2 && p->slice != q.slice) continue;      // not across slice boundaries
                        }
                        int qpav = (p->qp + q.qp + 1) >> 1;
                        int qpcav = (x264o_chroma_qp[clampi(p->qp + d.chroma_qp_offset, 0, 51)] + x264o_chroma_qp[clampi(q.qp + d.chroma_qp_offset, 0, 51)] + 1) >> 1;
                        int ia = clampi(qpav + alpha_off, 0, 51), ib = clampi(qpav + beta_off, 0, 51);
                        int ica = clampi(qpcav + alpha_off, 0, 51), icb = clampi(qpcav + beta_off, 0, 51);
                        for (int k = 0; k < 4; k++) {
                            int qb = vert ? kIdxOf[k][e] : kIdxOf[e][k];
                            int pb = vert ? kIdxOf[k][(e + 3) & 3] : kIdxOf[(e + 3) & 3][k];
                            int bs = bs_of(*p, pb, q, qb, e == 0);
                            if (!bs) continue;
                            int x = mbx * 16 + (vert ? e * 4 : k * 4), y = mby * 16 + (vert ? k * 4 : e * 4);
                            int tc0 = bs < 4 ? x264o_tc0_table[ia][bs - 1] : 0;
                            if (vert) x264o_deblock_luma_edge(Yp + (size_t)y * d.stride + x, 1, d.stride, 4, x264o_alpha_table[ia], x264o_beta_table[ib], tc0, bs);
                            else x264o_deblock_luma_edge(Yp + (size_t)y * d.stride + x, d.stride, 1, 4, x264o_alpha_table[ia], x264o_beta_table[ib], tc0, bs);
                            if (e & 1) continue;
                            int ctc0 = bs < 4 ? x264o_tc0_table[ica][bs - 1] : 0;
                            for (int c = 0; c < 2; c++) {
                                pixel *pc = UVp + (size_t)(y / 2) * d.stride + 2 * (x / 2) + c;
                                if (vert) x264o_deblock_chroma_edge(pc, 2, d.stride, 2, x264o_alpha_table[ica], x264o_beta_table[icb], ctc0, bs);
                                else x264o_deblock_chroma_edge(pc, d.stride, 2, 2, x264o_alpha_table[ica], x264o_beta_table[icb], ctc0, bs);
                            }
                        }
                    }
            }
    }
};

void finish_picture(Decoder &d)
{
    // output (cropped I420), then make the picture a reference: half-pel planes + borders
    int w = d.width, h = d.height;
    std::vector<uint8_t> f((size_t)w * h * 3 / 2);
    for (int y = 0; y < h; y++) memcpy(&f[(size_t)y * w], d.Y(d.cur) + (size_t)y * d.stride, w);
    uint8_t *u = &f[(size_t)w * h], *v = u + (size_t)(w / 2) * (h / 2);
    const pixel *uv = d.UV(d.cur);
    for (int y = 0; y < h / 2; y++) for (int x = 0; x < w / 2; x++) { u[y * (w / 2) + x] = uv[(size_t)y * d.stride + 2 * x]; v[y * (w / 2) + x] = uv[(size_t)y * d.stride + 2 * x + 1]; }
    d.frames.push_back(std::move(f));
    d.frame_pocs.push_back(d.cur_poc);
    d.have++;
    if (!d.cur_is_ref) return;                              // a non-reference picture leaves the DPB as it is
    pixel *planes[4] = { d.Y(d.cur, 0), d.Y(d.cur, 1), d.Y(d.cur, 2), d.Y(d.cur, 3) };
    x264o_frame_filter(planes, d.stride, d.mbw * 16, d.mbh * 16, d.pad);
    pixel *c = d.UV(d.cur);
    int cw = d.mbw * 8, ch = d.mbh * 8;
    for (int y = -d.cpad; y < ch + d.cpad; y++)
        for (int x = -d.cpad; x < cw + d.cpad; x++)
            if (x < 0 || x >= cw || y < 0 || y >= ch) {
                int sx = clampi(x, 0, cw - 1), sy = clampi(y, 0, ch - 1);
                c[(size_t)y * d.stride + 2 * x] = c[(size_t)sy * d.stride + 2 * sx];
                c[(size_t)y * d.stride + 2 * x + 1] = c[(size_t)sy * d.stride + 2 * sx + 1];
            }
    d.slot_mb[d.cur] = d.mb;
    for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) d.slot_lpoc[d.cur][l][i] = d.list_poc[l][i];
    // decoded reference picture marking (8.2.5): the slice's operations, else the sliding window
    const int max_frame_num = 1 << d.log2_max_frame_num;
    auto picnum = [&](const Decoder::Ref &r) { return r.frame_num > d.cur_frame_num ? r.frame_num - max_frame_num : r.frame_num; };
    if (!d.pending_mmco.empty()) {
        for (int pn : d.pending_mmco)
            for (size_t i = 0; i < d.dpb.size(); i++) if (picnum(d.dpb[i]) == (pn > d.cur_frame_num ? pn - max_frame_num : pn)) { d.dpb.erase(d.dpb.begin() + (long)i); break; }
    }
    if ((int)d.dpb.size() >= d.num_ref_frames) {
        size_t oldest = 0;
        for (size_t i = 1; i < d.dpb.size(); i++) if (picnum(d.dpb[i]) < picnum(d.dpb[oldest])) oldest = i;
        d.dpb.erase(d.dpb.begin() + (long)oldest);
    }
    d.dpb.push_back(Decoder::Ref{ d.cur, d.cur_frame_num, d.cur_poc });
}

static bool reject(int line) { if (getenv("X264O_DEC_DEBUG")) fprintf(stderr, "h264dec: rejected at line %d\n", line); return false; }

bool decode_nal(Decoder &d, const uint8_t *nal, size_t n)
{
    if (n < 2) return true;
    int type = nal[0] & 31;
    std::vector<uint8_t> rbsp;
    for (size_t i = 1; i < n; i++) {
        if (i >= 2 && nal[i] == 3 && nal[i - 1] == 0 && nal[i - 2] == 0) continue;   // emulation_prevention_three_byte
        rbsp.push_back(nal[i]);
    }
    BitReader br{ rbsp.data(), rbsp.size() };
    if (type == 7) {
        int profile = (int)br.get(8); br.get(8); br.get(8); br.ue();
        if (profile >= 100) {                               // High: 4:2:0, 8 bit, flat scaling lists only
            if (br.ue() != 1 || br.ue() != 0 || br.ue() != 0 || br.get1() || br.get1()) return reject(__LINE__);
        }
        d.log2_max_frame_num = (int)br.ue() + 4;
        d.poc_type = (int)br.ue();
        if (d.poc_type != 2 && d.poc_type != 0) return reject(__LINE__);
        if (d.poc_type == 0) d.log2_max_poc_lsb = (int)br.ue() + 4;
        d.num_ref_frames = (int)br.ue(); br.get1();
        if (d.num_ref_frames < 1 || d.num_ref_frames > 6) return reject(__LINE__);
        d.mbw = (int)br.ue() + 1; d.mbh = (int)br.ue() + 1;
        if (!br.get1()) return reject(__LINE__);                       // frame_mbs_only
        br.get1();
        d.crop_r = d.crop_b = 0;
        if (br.get1()) { br.ue(); d.crop_r = 2 * (int)br.ue(); br.ue(); d.crop_b = 2 * (int)br.ue(); }
        d.width = d.mbw * 16 - d.crop_r; d.height = d.mbh * 16 - d.crop_b;
        d.alloc();
        d.have_sps = !br.err;
        return d.have_sps;
    }
    if (type == 8) {
        br.ue(); br.ue();
        d.cabac = br.get1();                                // entropy_coding_mode_flag
        br.get1(); if (br.ue()) return reject(__LINE__);
        d.num_ref_default = (int)br.ue() + 1; d.num_ref1_default = (int)br.ue() + 1;
        d.weighted_pred = (int)br.get1();                              // weighted_pred_flag
        d.weighted_bipred_idc = (int)br.get(2);
        if (d.weighted_bipred_idc == 1) return reject(__LINE__);
        d.pic_init_qp = 26 + br.se(); br.se();
        d.chroma_qp_offset = br.se();
        d.deblock_ctrl = br.get1();
        if (br.get1()) return reject(__LINE__);                        // constrained intra
        br.get1();
        d.transform8x8_mode = 0;
        if (br.more_rbsp_data()) {
            d.transform8x8_mode = br.get1();
            if (br.get1()) return reject(__LINE__);                    // pic_scaling_matrix_present_flag
            if (br.se() != d.chroma_qp_offset) return reject(__LINE__); // second_chroma_qp_index_offset
        }
        d.have_pps = !br.err;
        return d.have_pps;
    }
    if (type == 1 || type == 5) {
        if (!d.have_sps || !d.have_pps) return reject(__LINE__);
        const int first_mb = (int)br.ue();                  // first_mb_in_slice: slices arrive in order and tile the picture
        if (first_mb != d.next_mb || first_mb >= d.mbw * d.mbh) return reject(__LINE__);
        int st = (int)br.ue() % 5;
        if (st != 0 && st != 1 && st != 2) return reject(__LINE__);
        br.ue();
        const int frame_num = (int)br.get(d.log2_max_frame_num), max_frame_num = 1 << d.log2_max_frame_num;
        if (type == 5) br.ue();
        const int nal_ref_idc = (nal[0] >> 5) & 3;
        int poc = 0;
        if (d.poc_type == 0) {
            // 8.2.1.1: PicOrderCntMsb from the previous REFERENCE picture's
            const int lsb = (int)br.get(d.log2_max_poc_lsb), max_lsb = 1 << d.log2_max_poc_lsb;
            if (type == 5) { d.prev_poc_msb = 0; d.prev_poc_lsb = 0; }
            int msb = d.prev_poc_msb;
            if (lsb < d.prev_poc_lsb && d.prev_poc_lsb - lsb >= max_lsb / 2) msb += max_lsb;
            else if (lsb > d.prev_poc_lsb && lsb - d.prev_poc_lsb > max_lsb / 2) msb -= max_lsb;
            poc = msb + lsb;
            if (first_mb == 0 && nal_ref_idc) { d.prev_poc_msb = msb; d.prev_poc_lsb = lsb; }
        } else poc = 2 * (type == 5 ? 0 : d.have);          // type 2: output order is decoding order (only used for bookkeeping here)
        if (first_mb == 0) {
            if (type == 5) { d.have = 0; d.dpb.clear(); }   // IDR empties the DPB
            d.cur_poc = poc; d.cur_frame_num = frame_num; d.cur_is_ref = nal_ref_idc != 0;
            // a free picture slot
            for (int sl = 0; sl < d.slots; sl++) { bool used = false; for (auto &r : d.dpb) used |= r.slot == sl; if (!used) { d.cur = sl; break; } }
            d.pending_mmco.clear();
        }
        if (st == 1) d.direct_spatial = br.get1();                     // direct_spatial_mv_pred_flag
        d.nref_active = d.num_ref_default; d.nref1_active = st == 1 ? d.num_ref1_default : 0;
        if (st != 2) {
            if (br.get1()) { d.nref_active = (int)br.ue() + 1; if (st == 1) d.nref1_active = (int)br.ue() + 1; }   // num_ref_idx_active_override_flag
            // ---- reference picture lists: initialisation (8.2.4.2) ----
            auto picnum = [&](const Decoder::Ref &r) { return r.frame_num > frame_num ? r.frame_num - max_frame_num : r.frame_num; };
            std::vector<Decoder::Ref> init[2];
            if (st == 0) {
                init[0] = d.dpb;
                std::sort(init[0].begin(), init[0].end(), [&](const Decoder::Ref &x, const Decoder::Ref &y) { return picnum(x) > picnum(y); });
            } else {
                std::vector<Decoder::Ref> before, after;
                for (auto &r : d.dpb) (r.poc < poc ? before : after).push_back(r);
                std::sort(before.begin(), before.end(), [](const Decoder::Ref &x, const Decoder::Ref &y) { return x.poc > y.poc; });
                std::sort(after.begin(), after.end(), [](const Decoder::Ref &x, const Decoder::Ref &y) { return x.poc < y.poc; });
                init[0] = before; init[0].insert(init[0].end(), after.begin(), after.end());
                init[1] = after; init[1].insert(init[1].end(), before.begin(), before.end());
                if (init[1].size() > 1 && init[0].size() == init[1].size()) {
                    bool same = true;
                    for (size_t i = 0; i < init[0].size(); i++) same &= init[0][i].slot == init[1][i].slot;
                    if (same) std::swap(init[1][0], init[1][1]);
                }
            }
            for (int l = 0; l <= (st == 1 ? 1 : 0); l++) {
                const int nact = l ? d.nref1_active : d.nref_active;
                std::vector<Decoder::Ref> list = init[l];
                if ((int)list.size() > nact) list.resize((size_t)nact);
                // ---- modification (8.2.4.3.1): short-term pictures by picture number difference ----
                if (br.get1()) {
                    int pred = frame_num, idx = 0;
                    for (;;) {
                        const int idc = (int)br.ue();
                        if (idc == 3) break;
                        if (idc > 1 || br.err) return reject(__LINE__);
                        const int absdiff = (int)br.ue() + 1;
                        int nowrap = idc == 0 ? pred - absdiff : pred + absdiff;
                        if (idc == 0 && nowrap < 0) nowrap += max_frame_num;
                        if (idc == 1 && nowrap >= max_frame_num) nowrap -= max_frame_num;
                        pred = nowrap;
                        const int pn = nowrap > frame_num ? nowrap - max_frame_num : nowrap;
                        const Decoder::Ref *pic = nullptr;
                        for (auto &r : d.dpb) if (picnum(r) == pn) pic = &r;
                        if (!pic) return reject(__LINE__);
                        // insert at idx, drop the later duplicate
                        const Decoder::Ref ins = *pic;
                        list.insert(list.begin() + idx, ins);
                        for (size_t j = (size_t)idx + 1; j < list.size(); j++) if (list[j].slot == ins.slot) { list.erase(list.begin() + (long)j); break; }
                        if ((int)list.size() > nact) list.resize((size_t)nact);
                        idx++;
                    }
                }
                if ((int)list.size() < nact) return reject(__LINE__);  // refers to pictures not in the DPB
                for (int i = 0; i < nact; i++) { d.list_slot[l][i] = list[(size_t)i].slot; d.list_poc[l][i] = list[(size_t)i].poc; }
            }
        }
        if (d.weighted_pred && st == 0) {                   // pred_weight_table()
            d.luma_logwd = (int)br.ue(); d.chroma_logwd = (int)br.ue();
            if (d.luma_logwd > 7 || d.chroma_logwd > 7) return reject(__LINE__);
            bool any_l = false, any_c = false;
            for (int i = 0; i < d.nref_active; i++) {
                d.wp[i].lw = 1 << d.luma_logwd; d.wp[i].lo = 0;
                d.wp[i].cw[0] = d.wp[i].cw[1] = 1 << d.chroma_logwd; d.wp[i].co[0] = d.wp[i].co[1] = 0;
                if (br.get1()) {
                    d.wp[i].lw = br.se(); d.wp[i].lo = br.se(); any_l = true;
                    if (d.wp[i].lw < -128 || d.wp[i].lw > 127 || d.wp[i].lo < -128 || d.wp[i].lo > 127) return reject(__LINE__);      // (an inferred weight is 2^logWD: up to 128)
                }
                if (br.get1()) { for (int c = 0; c < 2; c++) { d.wp[i].cw[c] = br.se(); d.wp[i].co[c] = br.se(); } any_c = true; }
            }
            d.weighted_slices[0] += any_l; d.weighted_slices[1] += any_c;
        }
        if (nal_ref_idc) {
            if (type == 5) { br.get1(); br.get1(); }
            else if (br.get1()) {                           // adaptive_ref_pic_marking_mode_flag
                for (;;) {
                    const int op = (int)br.ue();
                    if (op == 0) break;
                    if (op != 1 || br.err) return reject(__LINE__);    // only "mark a short-term picture unused"
                    const int pn = frame_num - ((int)br.ue() + 1);
                    if (first_mb == 0) d.pending_mmco.push_back(pn);
                }
            }
        }
        if (d.cabac && st != 2 && br.ue() != 0) return reject(__LINE__);   // cabac_init_idc: only the tables of 0 are in this checker
        int qp = d.pic_init_qp + br.se();
        int disable = 0, a = 0, b = 0;
        if (d.deblock_ctrl) { disable = (int)br.ue(); if (disable != 1) { a = 2 * br.se(); b = 2 * br.se(); } }
        if (getenv("X264O_DEC_DEBUG")) fprintf(stderr, "slice: type %d frame_num %d poc %d nref %d/%d qp %d disable %d bitpos %zu l0 %d,%d,%d l1 %d,%d\n", st, d.cur_frame_num, d.cur_poc, d.nref_active, d.nref1_active, qp, disable, br.pos,
                                               d.list_poc[0][0], d.list_poc[0][1], d.list_poc[0][2], d.list_poc[1][0], d.list_poc[1][1]);
        SliceDec sd{ d, br, st, qp, disable, a, b };
        sd.first_mb = first_mb; sd.slice_no = d.slice_no++;
        if (first_mb == 0) d.mb_bits.resize(d.mb_bits.size() + (size_t)(d.mbw * d.mbh), 0);
        if (first_mb) { if (disable != d.pic_disable || a != d.pic_a || b != d.pic_b) return reject(__LINE__); }      // one filter setting per picture in this checker
        else { d.pic_disable = disable; d.pic_a = a; d.pic_b = b; }
        sd.run();
        if (br.err || sd.end_mb <= first_mb) return reject(__LINE__);
        d.next_mb = sd.end_mb;
        if (d.next_mb == d.mbw * d.mbh) {                   // the picture is complete
            sd.deblock();
            finish_picture(d);
            d.next_mb = 0; d.slice_no = 0;
        }
        return true;
    }
    return true;     // SEI, AUD, ... ignored
}

}  // namespace

static std::vector<int> g_mb_bits;       // per-macroblock CAVLC bit counts of the last x264o_h264_decode call

extern "C" {

// after x264o_h264_decode of a CAVLC stream: bits the macroblock layer of every macroblock took (pictures back to back, 0 = skipped)
int x264o_h264_last_mb_bits(int *out, int cap)
{
    const int n = (int)g_mb_bits.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = g_mb_bits[(size_t)i];
    return n;
}

static std::vector<int> g_pocs;
static int g_weighted[2];
// P slices of the last x264o_h264_decode call that carried an explicit luma weight (out[0]) / chroma weights (out[1])
extern "C" void x264o_h264_last_weighted(int *out) { out[0] = g_weighted[0]; out[1] = g_weighted[1]; }
// POC of every picture of the last x264o_h264_decode call, in decoding order (the pictures are returned in that order)
int x264o_h264_last_pocs(int *out, int cap)
{
    const int n = (int)g_pocs.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = g_pocs[(size_t)i];
    return n;
}

// Decodes an Annex-B stream.  Returns the number of pictures, or -1 on a syntax error / unsupported
// feature.  Pictures are written back to back (cropped I420) into `out` if it is large enough.
int x264o_h264_decode(const uint8_t *data, size_t n, uint8_t *out, size_t out_cap, int *width, int *height)
{
    Decoder d;
    size_t i = 0;
    auto is_start = [&](size_t k) { return k + 2 < n && data[k] == 0 && data[k + 1] == 0 && data[k + 2] == 1; };
    while (i < n && !is_start(i)) i++;
    while (i < n) {
        size_t s = i + 3, e = s;
        while (e < n && !is_start(e)) e++;
        size_t end = e;
        while (end > s && data[end - 1] == 0 && e < n) end--;       // trailing_zero_8bits / 4-byte start codes
        if (!decode_nal(d, data + s, end - s)) return -1;
        i = e;
    }
    if (width) *width = d.width;
    if (height) *height = d.height;
    g_mb_bits = d.mb_bits; g_pocs = d.frame_pocs; g_weighted[0] = d.weighted_slices[0]; g_weighted[1] = d.weighted_slices[1];
    size_t fsz = (size_t)d.width * d.height * 3 / 2, off = 0;
    for (auto &f : d.frames) { if (off + fsz <= out_cap) memcpy(out + off, f.data(), fsz); off += fsz; }
    return (int)d.frames.size();
}

}

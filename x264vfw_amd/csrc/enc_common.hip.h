// enc_common.hip.h — device-side view of one lock-step batch of streams (frame pipeline, Tier 2).
#pragma once
#include "common.hip.h"
#include "mc.hip.h"
#include "intra.hip.h"

namespace x264gpu {

constexpr int PAD = 32;    // luma padding of reference planes (samples)
constexpr int CPAD = 16;   // chroma padding (chroma samples; 2*CPAD bytes in NV12)
constexpr int MVCOST_HALF = 32768;

// Everything a pipeline kernel needs, passed by value.  All per-stream buffers are [streams][...] with
// the per-stream byte strides below, so blockIdx.y (or the workgroup index) selects the stream.
struct EncK {
    int w, h;                 // picture size
    int mbw, mbh, nmb;        // macroblock grid
    int cw, ch;               // coded size (multiples of 16)
    int fs, rs;               // fenc stride, reference-plane stride (bytes)
    size_t fency_bytes, fencuv_bytes;        // per stream
    size_t plane_bytes, luma_bytes;          // one padded plane; 4 planes
    size_t cplane_bytes;                     // padded NV12 plane
    const uint8_t *i420;      // [streams] tightly packed input pictures
    uint8_t *fenc_y, *fenc_uv;
    uint8_t *rec_luma, *rec_chroma;          // DPB slot being reconstructed
    // DPB slots of the slice's references, list 0 then list 1 in ONE index space: list 0 index r = entry r (0 = nearest), list 1 index r (B slices)
    // = entry nref + r — motion search, reference cache and vector side data only ever see this combined index
    const uint8_t *ref_luma[8], *ref_chroma[8];
    int nref;                                // references of list 0 (all a P slice has)
    int nref1;                               // references of list 1 (B slices)
    // P slices, x264 --weightp: explicit luma weight of list-0 index r packed as offset (int8) | scale (int8) << 8 | denom << 16 | on << 24
    // (8.4.2.3.2; applied AFTER the quarter-pel interpolation, as mc.get_ref / mc_luma do) and the index of x264's blind duplicate of reference 0
    // (h->mb.ref_blind_dupe; 0 = none): the same picture as index 0, refined from index 0's vector instead of searched
    int wl0[8], blind_dupe, wp_any;
    int wc0[16], wc_any;      // explicit chroma weights of list-0 index r: wc0[2 r] Cb, [2 r + 1] Cr, packed like wl0; wc_any: some index has one
    int refpic[8];                           // the picture behind list-0 index r as an index without duplicates (order of first appearance): reference cache tags, loop filter
    uint8_t biw[5][4];                       // B: implicit bi-prediction weight of the list-0 sample for (list-0 index, list-1 index), of 64 (x264 bipred_weight)
    const int8_t *colref; const int16_t *colmv;    // B: per 8x8 block of the first picture of list 1, the reference index it used (-1 intra) and that vector
    int8_t *colref_cur; int16_t *colmv_cur;  // the same of the picture being coded, for the B pictures that will have it at the head of their list 1
    // --direct temporal / auto (B slices): the co-located picture's own list-0 index per 8x8 block (-1: none / intra) and its macroblock types, this
    // picture's copy of the former; per stream 1 = temporal direct prediction (nullptr: spatial everywhere); direct_auto: both modes are probed and
    // counted into dscore[streams][2] ([0] temporal, [1] spatial); x264's map_col_to_list0 and dist_scale_factor[r][0]
    const int8_t *colref0; const uint8_t *coltype; int8_t *colref0_cur;
    const uint8_t *direct_flags; int direct_auto; int *dscore;
    int map_col[8], dist_scale[8];
    const int16_t *lowres_mv1;               // B: lookahead vectors towards the first picture of list 1
    x264gpu_mb *mb;           // [streams][nmb]
    int16_t *levels;          // [streams][nmb][416]
    int qp, lambda, qpc;
    int me_range, subme, dct_decimate, partitions, chroma_qp_offset;
    int slice_type;
    int alpha_off, beta_off;  // deblock offsets (already *2)
    int deblock_rdo;          // x264 b_deblock_rdo (--subme 9 and up, cfg.rd bit 6, loop filter on): whole-macroblock RD candidates are measured after x264_macroblock_deblock
    Q4 q_luma_intra, q_luma_inter, q_chroma_intra, q_chroma_inter;
    Q8 q8_intra, q8_inter;    // 8x8 luma transform (dct8x8)
    int dct8x8;
    int me_method;            // 0 dia, 1 hex, 2 umh, 3 esa
    int chroma_me;            // sub-pel SATD costs carry chroma (subme >= 5)
    int mixed_refs;           // 8x8 blocks / 16x8, 8x16 halves pick their own reference
    unsigned long long *dbg;  // optional diagnostics (NULL in production): per (stream, wave) cycle counters
    // adaptive quantisation (aq_mode 1): per-macroblock quantisers and the tables every quantiser-dependent value is read from
    uint8_t *mbqp;            // [streams][nmb]; NULL = every macroblock uses the slice quantiser (the scalars above)
    const Q4 *q4tab;          // [52][4]: luma intra, luma inter, chroma intra, chroma inter
    const Q8 *q8tab;          // [52][2]: intra, inter
    const int *lambda_tab;    // [52]
    const uint16_t *cost_all; // [52][2 * MVCOST_HALF]
    const int8_t *stream_qp;  // optional [streams]: each stream's slice quantiser (x264gpu_encoder_set_stream_qps) and
    const float *stream_qpm;  // ... [streams] its float quantiser (x264gpu_pic.qpm of that stream; never 0 here); k.qp / k.qpm otherwise
    float aq_strength;        // x264_adaptive_quant_frame's strength of mode 1 (aq-strength * 1.0397f)
    float qpm;                // the picture's float quantiser (x264 rc->qpm; (float)qp in constant-quantiser sessions): enters the per-macroblock quantisers before the rounding
    int qp_snap;              // --aq-mode != 0: a macroblock quantiser within 1 of the previous macroblock's takes that one (x264_macroblock_analyse)
    int *wf_progress;         // [streams][2][WFG_ROWS]: row counters of the wavefront kernels when ONE stream spans several workgroups
    // motion side data of the raster macroblock loop (k_mb.hip.h; x264: h->mb.mvr, frame->mv16x16, frame->mb_type)
    int16_t *mv16_cur;        // [streams][nmb][2]: 16x16 search result in reference 0 of the picture being coded (= mvr[0]); lives with the DPB slot
    const int16_t *mv16_ref0; // the same array of reference 0 (temporal candidates)
    int16_t *mvr[8];          // [combined index >= 1][streams][nmb][2]: 16x16 search results per reference (list 0 index 0 lives in mv16_cur)
    uint8_t *mbtype_cur;      // [streams][nmb] macroblock types of the picture being coded; lives with the DPB slot
    const uint8_t *mbtype_ref0;
    int tscale[8];            // (POC distance to reference r) * inv_ref_poc of reference 0, for the temporal candidates (combined index)
    int temporal;             // reference 0 was itself a P picture: its 16x16 vectors are search candidates
    const int16_t *lowres_mv; // optional [streams][nmb][2] lookahead vectors (x264 lowres_mvs[0][0]); first entry 0x7fff = absent
    int fast_pskip, mv_range;
    int rd, psy, psy_rd_q8;   // RD mode decision (subme 6 / 7 of a CAVLC session), b_psy, FIX8(psy-rd strength)
    uint8_t *tc, *amvd; uint32_t *cab_out;
    // --slices N (slices_plain), P pictures: x264 codes the slices one after the other and its fast-intra decision reads the number of intra
    // macroblocks of the picture so far.  The slices run side by side here on an ASSUMED count of the slices before them (sl_stat[..][3]) and
    // report their own count plus the window of counts [hi, lo) for which every decision they took stays what it is ([0], [1], [2]); slices
    // whose window misses the real sum run again (sl_rerun), until none does (k_slice_priors, encoder.hip)
    int *sl_stat, *sl_rerun; int sl_pass;
    int trellis; const uint16_t *tr_su; const uint8_t *tr_tu; const int *tr_l2;      // trellis sites of the final encode (mask) + x264_rdo_init's tables              // RD: [streams][nmb][24] total_coeff of every block of the picture being coded (nC of the bit counts)
    // load balance of the lock-step batch (encoder.hip k_balance): perm[workgroup] = the stream it codes (null: identity), wtime[stream] = cycles its
    // wavefront took (written by the macroblock loop, read for the next picture of the same kind)
    const int *perm; unsigned *wtime;
    const uint32_t *ctab;     // CABAC sessions with RD: the chain table of the size-only coder (cabac_rd.hip.h cab_chain; prim_kernels.hip cabac_chain_table)
    int cabac;                // the session's entropy coder is CABAC: P8x8 cost details of x264's analysis depend on it
    int slices;               // x264 slice threads: slices per picture (rows split evenly), 1 = one
    int slices_plain;         // x264 --slices N rather than slice threads: the loop filter crosses slice boundaries
    unsigned long long *prof; // MB_PROF builds only: [streams][16] cycle counters of the macroblock loop's phases (null otherwise)
};
// the slice quantiser of stream s
__device__ __forceinline__ int slice_qp(const EncK &k, int s) { return k.stream_qp ? (int)k.stream_qp[s] : k.qp; }
// ... the stream's float quantiser (rate-controlled sessions; x264 rc->qpm)
__device__ __forceinline__ float slice_qpm(const EncK &k, int s) { return k.stream_qp ? k.stream_qpm[s] : k.qpm; }
// x264's single-float helpers (common/common.h x264_log2, x264_exp2fix8; ratecontrol.c x264_ratecontrol_mb_qp), evaluated as the C source reads: every
// product and sum rounded on its own (no fused multiply-add), so that the host checker's -ffp-contract=off build and the device agree to the bit
// (the compiler's __fmul_rn / __fadd_rn are plain operators that it may fuse; these carry no contraction flag into whatever they are inlined into)
__device__ __forceinline__ float f_mul(float a, float b) {
#pragma clang fp contract(off)
    return a * b; }
__device__ __forceinline__ float f_add(float a, float b) {
#pragma clang fp contract(off)
    return a + b; }
__device__ __forceinline__ float f_sub(float a, float b) {
#pragma clang fp contract(off)
    return a - b; }
__device__ __forceinline__ float f_div(float a, float b) {
#pragma clang fp contract(off)
    return a / b; }
static __constant__ float c_x264_log2_lut[128] = {
#include "x264gpu_log2f_lut.inc"
};
static __constant__ uint16_t c_x264_exp2_lut[64] = {
#include "x264gpu_exp2_lut.inc"
};
__device__ __forceinline__ float x264_log2(unsigned x) { const int lz = __builtin_clz(x); return f_add(c_x264_log2_lut[((x << lz) >> 24) & 0x7f], (float)(31 - lz)); }
__device__ __forceinline__ int x264_exp2fix8(float x)
{
    const int i = (int)f_add(f_mul(x, -64.f / 6.f), 512.5f);
    if (i < 0) return 0;
    if (i > 1023) return 0xffff;
    return (int)(((unsigned)(c_x264_exp2_lut[i & 63] + 256) << (i >> 6)) >> 8);
}
// clip3((int)(qpm + offset + 0.5f), 1, 51)
__device__ __forceinline__ int x264_mb_qp(float qpm, float offset) { return min(max((int)f_add(f_add(qpm, offset), 0.5f), 1), 51); }
// mbtree_propagate_cost (common/mc.c), one block; fps_factor = 1 / 512 (constant frame rate, MBTREE_PRECISION 0.5f)
__device__ __forceinline__ int x264_propagate_amount(int propagate_in, int intra_cost, int inter_cost, int inv_qscale)
{
    if (!intra_cost) return 0;
    const float propagate_intra = (float)(intra_cost * inv_qscale);
    const float propagate_amount = f_add((float)propagate_in, f_mul(propagate_intra, 1.f / 512.f));
    const float propagate_num = (float)(intra_cost - inter_cost), propagate_denom = (float)intra_cost;
    return min((int)f_add(f_div(f_mul(propagate_amount, propagate_num), propagate_denom), 0.5f), 32767);
}


__device__ __forceinline__ const uint8_t *ref_plane00(const EncK &k, int s, int r)
{
    return k.ref_luma[r] + (size_t)s * k.luma_bytes + (size_t)PAD * k.rs + PAD;
}
__device__ __forceinline__ uint8_t *rec_plane00(const EncK &k, int s)
{
    return k.rec_luma + (size_t)s * k.luma_bytes + (size_t)PAD * k.rs + PAD;
}
// explicit weighted prediction of four luma samples (x264 mc_weight: opscale / opscale_noden), wpk = EncK::wl0[r]
__device__ __forceinline__ uint32_t wp4(uint32_t p, int wpk)
{
    const int offset = (int)(int8_t)(wpk & 0xff), scale = (int)(int8_t)((wpk >> 8) & 0xff), denom = (wpk >> 16) & 0xff;
    const int rnd = denom ? 1 << (denom - 1) : 0;
    int v[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { v[i] = (((int)((p >> (8 * i)) & 0xff) * scale + rnd) >> denom) + offset; v[i] = v[i] < 0 ? 0 : v[i] > 255 ? 255 : v[i]; }
    return (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
}
// the picture behind list-0 index r of a P slice, as an index without duplicates (reference cache tags, loop filter): the duplicate is picture 0
__device__ __forceinline__ int ref_picture(const EncK &k, int r) { return k.refpic[r]; }
__device__ __forceinline__ const uint8_t *ref_chroma00(const EncK &k, int s, int r)
{
    return k.ref_chroma[r] + (size_t)s * k.cplane_bytes + (size_t)CPAD * k.rs + 2 * CPAD;
}
// bits of ref_idx te(v) with nref active references
// x264 slice threads split the macroblock rows evenly: does a slice (other than the first) begin at row mby
__device__ __forceinline__ bool slice_starts_at_row(const EncK &k, int mby)
{
    for (int sl = 1; sl < k.slices; sl++) if ((k.mbh * sl + k.slices / 2) / k.slices == mby) return true;
    return false;
}
// the loop filter leaves the top edge of row mby alone: slice threads code disable_deblocking_filter_idc 2, plain --slices N code 0
__device__ __forceinline__ bool filter_stops_at_row(const EncK &k, int mby) { return !k.slices_plain && slice_starts_at_row(k, mby); }
__device__ __forceinline__ int ref_bits(int nref, int r) { return nref <= 1 ? 0 : nref == 2 ? 1 : 2 * (31 - __builtin_clz(r + 1)) + 1; }
__device__ __forceinline__ uint8_t *rec_chroma00(const EncK &k, int s)
{
    return k.rec_chroma + (size_t)s * k.cplane_bytes + (size_t)CPAD * k.rs + 2 * CPAD;
}

static __device__ const uint8_t d_chroma_qp_table[52] = {
    0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
    18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
    34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };
__device__ __forceinline__ int chroma_qp_dev(int q) { return d_chroma_qp_table[q < 0 ? 0 : q > 51 ? 51 : q]; }

__device__ __forceinline__ int bs_size_ue(int v)
{
    return 2 * (31 - __builtin_clz(v + 1)) + 1;
}

// ---- shared residual coders (used by the inter and the intra kernels) ------------------------------

// scatter this lane's 4 levels (natural layout: row j, columns 0..3) to scan order in `dst[16]`
// scan (zigzag) index of the four coefficients a lane holds in natural layout (row j, columns 0..3),
// packed one nibble each: row 0 -> {0,1,5,6}, row 1 -> {2,4,7,12}, row 2 -> {3,8,11,13}, row 3 -> {9,10,14,15}
__device__ __forceinline__ unsigned scan_nibbles(int j)
{
    return j == 0 ? 0x6510u : j == 1 ? 0xC742u : j == 2 ? 0xDB83u : 0xFEA9u;
}
__device__ __forceinline__ void store_levels_scan(int16_t *dst, const int v[4], int j)
{
    const unsigned z = scan_nibbles(j);
    dst[z & 15] = (int16_t)v[0]; dst[(z >> 4) & 15] = (int16_t)v[1];
    dst[(z >> 8) & 15] = (int16_t)v[2]; dst[z >> 12] = (int16_t)v[3];
}
__device__ __forceinline__ unsigned scan_mask(const int v[4], int j)
{
    const unsigned z = scan_nibbles(j);
    return ((v[0] != 0 ? 1u : 0u) << (z & 15)) | ((v[1] != 0 ? 1u : 0u) << ((z >> 4) & 15)) |
           ((v[2] != 0 ? 1u : 0u) << ((z >> 8) & 15)) | ((v[3] != 0 ? 1u : 0u) << (z >> 12));
}
__device__ __forceinline__ bool any_big(const int v[4])
{
    return (int)(abs(v[0]) > 1) | (int)(abs(v[1]) > 1) | (int)(abs(v[2]) > 1) | (int)(abs(v[3]) > 1);      // branch-free on purpose
}

// gather 4 chroma samples of one plane (c = 0 U, 1 V) from 8 interleaved NV12 bytes
__device__ __forceinline__ uint32_t nv12_pick(uint32_t lo, uint32_t hi, int c)
{
    lo >>= 8 * c; hi >>= 8 * c;
    return (lo & 0xff) | ((lo >> 8) & 0xff00) | ((hi & 0xff) << 16) | ((hi << 8) & 0xff000000u);
}

// Chroma residual of one macroblock.  Lanes 0..31 active: plane c = lane>>4 (one DPP row per plane),
// 4x4 block i = (lane>>2)&3, row j = lane&3.  enc/pred are this lane's 4 samples (packed).  Returns the
// reconstructed 4 samples; writes levels + updates nnz / cbp via the out-params (valid on every lane).
__device__ __forceinline__ uint32_t chroma_residual(uint32_t enc, uint32_t pred, const Q4 &q, bool inter, bool decimate,
                                                    int lane, int16_t *lv, unsigned &nnz_bits, int &cbp_chroma)
{
    const int c = (lane >> 4) & 1, i = (lane >> 2) & 3, j = lane & 3;
    const bool act = lane < 32;
    int e[4], p[4], v[4];
    unpack4(enc, e); unpack4(pred, p);
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = act ? e[t] - p[t] : 0;
    dct4_quad(v, lane);
    // DC of each block sits at (row 0, col 0): collect the plane's four DCs on every lane of the plane
    int dcs[4];
#pragma unroll
    for (int b = 0; b < 4; b++) dcs[b] = __shfl(v[0], (lane & 48) + 4 * b);
    if (j == 0) v[0] = 0;
    quant4_row(v, q, j);
    unsigned mask = quad_or((int)scan_mask(v, j));
    int big = quad_or(any_big(v) ? 1 : 0);
    bool nz = mask != 0;
    int score = 0;
    if (inter && decimate) {
        int s = nz ? (big ? 9 : decimate_from_mask(mask, 1)) : 0;
        score = row16_sum(j == 0 ? s : 0);
    }
    bool plane_ac = row16_or(nz ? 1 : 0) != 0;
    if (plane_ac && inter && decimate && score < 7) plane_ac = false;
    // 2x2 DC Hadamard + quant (every lane of the plane redundantly)
    int f[4];
    { int a = dcs[0] + dcs[1], b = dcs[0] - dcs[1], cc = dcs[2] + dcs[3], d = dcs[2] - dcs[3];
      f[0] = a + cc; f[1] = b + d; f[2] = a - cc; f[3] = b - d; }
    int ldc[4], nzdc = 0;
#pragma unroll
    for (int b = 0; b < 4; b++) { ldc[b] = quant_one(f[b], q.mf[0] >> 1, q.bias[0] << 1); nzdc |= ldc[b]; }
    // DC-only planes: x264_mb_optimize_chroma_dc (oracle x264o_optimize_chroma_2x2_dc) lowers the DC levels while the
    // reconstruction ((idct2x2(level) * dmf >> 5) + 32) >> 6 stays the same; visiting order 3, 1, 2, 0; quantisers whose
    // dmf exceeds 2048 are left alone.  Every lane of the plane runs the same few iterations.
    if (nzdc && !plane_ac) {
        const int dmf = q.dq[0] << (q.qp / 6);
        if (dmf <= 32 * 64) {
            auto rnd = [&](int o[4]) {
                const int d0 = ldc[0] + ldc[1], d1 = ldc[2] + ldc[3], d2 = ldc[0] - ldc[1], d3 = ldc[2] - ldc[3];
                o[0] = ((d0 + d1) * dmf >> 5) + 32; o[1] = ((d0 - d1) * dmf >> 5) + 32;
                o[2] = ((d2 + d3) * dmf >> 5) + 32; o[3] = ((d2 - d3) * dmf >> 5) + 32;
            };
            int ref[4], out[4];
            rnd(ref);
            if (!((ref[0] | ref[1] | ref[2] | ref[3]) >> 6)) { ldc[0] = ldc[1] = ldc[2] = ldc[3] = 0; nzdc = 0; }
            else {
                int left = 0;
#define X264GPU_OPT_DC(C) { int level = ldc[C]; const int sign = level >> 31 | 1; \
                    while (level) { ldc[C] = level - sign; rnd(out); \
                        if (((ref[0] ^ out[0]) | (ref[1] ^ out[1]) | (ref[2] ^ out[2]) | (ref[3] ^ out[3])) >> 6) { left = 1; ldc[C] = level; break; } \
                        level -= sign; } }
                X264GPU_OPT_DC(3) X264GPU_OPT_DC(1) X264GPU_OPT_DC(2) X264GPU_OPT_DC(0)
#undef X264GPU_OPT_DC
                if (!left) { ldc[0] = ldc[1] = ldc[2] = ldc[3] = 0; nzdc = 0; }
                else nzdc = 1;
            }
        }
    }
    int dq[4] = { 0, 0, 0, 0 };
    if (nzdc) {
        int a = ldc[0] + ldc[1], b = ldc[0] - ldc[1], cc = ldc[2] + ldc[3], d = ldc[2] - ldc[3];
        int g[4] = { a + cc, b + d, a - cc, b - d };
        int ls = q.dq[0] << (q.qp / 6);
#pragma unroll
        for (int b2 = 0; b2 < 4; b2++) dq[b2] = (g[b2] * ls) >> 5;
    }
    const bool keep = plane_ac && nz;
    if (act) {
        int16_t *l = lv + X264GPU_LV_CHROMA_AC + (c * 4 + i) * 16;
        int z[4] = { 0, 0, 0, 0 };
        store_levels_scan(l, keep ? v : z, j);
        if (i == 0 && j == 0)
#pragma unroll
            for (int b = 0; b < 4; b++) lv[X264GPU_LV_CHROMA_DC + c * 4 + b] = (int16_t)ldc[b];
    }
    if (keep) dequant4_row(v, q, j);
    else { v[0] = v[1] = v[2] = v[3] = 0; }
    if (j == 0) v[0] = dq[i];
    idct4_quad(v, lane);
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] += p[t];
    // flags: ballot over one lane per block
    unsigned long long bal = __ballot(act && keep && j == 0);
    unsigned long long bdc = __ballot(act && nzdc != 0 && i == 0 && j == 0);
    unsigned bits = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) bits |= (unsigned)((bal >> (4 * b)) & 1) << (16 + b);
    bits |= (unsigned)(bdc & 1) << 25;
    bits |= (unsigned)((bdc >> 16) & 1) << 26;
    nnz_bits |= bits;
    cbp_chroma = (bits & 0x00ff0000u) ? 2 : (bits & 0x06000000u) ? 1 : 0;
    return pack4_clip(v);
}

// ------------------------------------------------------------------------------------------------
// Row counters of the two wavefront kernels.  MWG = false: one workgroup owns the stream, counters live in LDS and hand-offs
// are workgroup-scope (same CU, coherent L1).  MWG = true: the rows of ONE stream are dealt to several workgroups (few streams
// in flight: single-stream latency), counters live in global memory, loads / stores are agent-scope atomics and the fences
// make the reconstructed pixels and records of the row above visible across CUs.
// ------------------------------------------------------------------------------------------------
constexpr int WFG_ROWS = 160;
template <bool MWG> __device__ __forceinline__ int wfp_load(const int *p)
{
    if (MWG) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *(const volatile int *)p;
}
template <bool MWG> __device__ __forceinline__ void wfp_store(int *p, int v)
{
    if (MWG) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(volatile int *)p = v;
}
template <bool MWG> __device__ __forceinline__ void wfp_release()
{
    if (MWG) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}
template <bool MWG> __device__ __forceinline__ void wfp_acquire()
{
    if (MWG) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

}  // namespace x264gpu

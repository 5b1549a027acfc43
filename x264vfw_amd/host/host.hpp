// host.hpp — internal declarations of the host-side encoder shell (boundary B1 behind include/x264.h).
#pragma once
#include "../../include/x264.h"
#include "../../include/x264gpu.h"
#include "bitstream.hpp"
#include <string.h>
#include <vector>

namespace x264host {

struct SliceParams {
    int mbw, mbh;
    int slice_type;          // X264GPU_SLICE_I / _P
    int qp, pic_init_qp;
    int frame_num, log2_max_frame_num;
    int idr, idr_pic_id, nal_ref_idc;
    int pps_id;
    int num_ref;             // active references of this slice (te() range)
    int num_ref_default;     // PPS num_ref_idx_l0_default_active
    int disable_deblock_idc, alpha_off_div2, beta_off_div2;
    int transform8x8_mode;   // PPS transform_8x8_mode_flag
    int cabac;               // PPS entropy_coding_mode_flag: CABAC slice data (cabac_init_idc 0), else CAVLC
    int slices_plain = 0;    // several slices per picture are x264's --slices N (the loop filter crosses their boundaries: idc stays 0) rather than slice threads (idc 2)
    int first_row = 0, end_row = 0;   // macroblock rows [first_row, end_row) of this slice (end_row 0 = mbh: one slice per picture); nothing
                             // above first_row is available to the slice's predictions (7.4.1.2.4 / 6.4.x availability)
    // ---- sessions with B pictures (x264 slice_header_write with sps->i_poc_type 0) ----
    int log2_max_poc_lsb = 0;    // > 0: pic_order_cnt_type 0, pic_order_cnt_lsb = poc mod 2^log2_max_poc_lsb is sent
    int poc = 0;
    int num_ref1 = 0, num_ref1_default = 1;      // B: active references of list 1, the PPS default
    int direct_spatial = 1;      // direct_spatial_mv_pred_flag
    // ref_pic_list_modification of list 0 / 1 (x264 b_ref_pic_list_reordering): the whole list as picture-number differences, when the encoder's
    // order (nearest in display order first) is not the default initial order of 8.2.4.2
    struct Reorder { int n = 0; struct { int idc, arg; } cmd[16]; } reorder[2];
    // memory_management_control_operation 1 (mark a short-term picture unused) x n_mmco: difference_of_pic_nums (x264's b-pyramid bookkeeping)
    int n_mmco = 0, mmco_diff[16] = { 0 };
    // pred_weight_table of P slices (x264 --weightp, PPS weighted_pred_flag): present whenever the PPS flag is set.  luma_log2_weight_denom /
    // chroma_log2_weight_denom = the denominator of the first weighted index (x264: all weighted indices share it), else 0; an index with one chroma
    // plane weighted sends the other as { 1 << denom, 0 } (x264 weighted_pred_init)
    int weighted_pred = 0;
    struct { int on, denom, scale, offset; } wl0[X264GPU_MAX_LIST] = {};
    struct { int on[2], denom, scale[2], offset[2]; } wc0[X264GPU_MAX_LIST] = {};
};
// what a slice took, split as x264's h->stat.frame does for the 2-pass statistics: the macroblock headers (types, prediction modes, references, vector
// differences: i_mv_bits) and what follows them (coded block pattern, quantiser delta, residual: i_tex_bits); everything else is 'misc'
struct SliceStats { int skip; long mv_bits = 0, tex_bits = 0; };

struct SpsParams {
    int profile_idc, level_idc, sps_id;
    int mbw, mbh, crop_right, crop_bottom;       // crop in luma samples
    int num_ref_frames, log2_max_frame_num;
    int sar_w, sar_h, fullrange, colorprim, transfer, colmatrix, overscan, vidformat;
    uint32_t num_units_in_tick, time_scale;
    int constraint_set0, constraint_set1;
    int mv_range;            // --mvrange (luma samples): log2_max_mv_length_* = floor(log2(4 * mv_range - 1)) + 1, as x264's sps init
    int log2_max_poc_lsb = 0;    // > 0: pic_order_cnt_type 0 (sessions with B pictures), else type 2
    int num_reorder_frames = 0;  // x264: 2 with --b-pyramid, 1 with B pictures, else 0
};
struct PpsParams { int pps_id, sps_id, cabac, num_ref, pic_init_qp, chroma_qp_offset, transform8x8_mode; int weighted_bipred_idc = 0; int weighted_pred = 0; };

void write_sps(std::vector<uint8_t> &out, const SpsParams &s, bool annexb);
void write_pps(std::vector<uint8_t> &out, const PpsParams &p, bool annexb);
void write_sei_version(std::vector<uint8_t> &out, const char *text, bool annexb);
void write_slice_header(BitWriter &bw, const SliceParams &p);
// the levels of macroblock i: levels + i * X264GPU_MB_LEVELS, or — index != nullptr: the device packed them (x264gpu_pack_levels) — its kept groups of 16 spread out
// into `scratch` (X264GPU_MB_LEVELS zeros first)
inline const int16_t *mb_levels(const int16_t *levels, const x264gpu_level_index *index, size_t i, int16_t *scratch)
{
    if (!index) return levels + i * X264GPU_MB_LEVELS;
    memset(scratch, 0, X264GPU_MB_LEVELS * sizeof(int16_t));
    const int16_t *src = levels + (size_t)index[i].at * 16;
    for (uint32_t g = index[i].groups; g; g &= g - 1, src += 16) memcpy(scratch + 16 * __builtin_ctz(g), src, 32);
    return scratch;
}
void write_slice(std::vector<uint8_t> &out, const SliceParams &p, const x264gpu_mb *mbs, const int16_t *levels,
                 bool annexb, bool long_startcode, SliceStats *stats, int threads = 1, const x264gpu_level_index *index = nullptr);
int slice_first_row(int mbh, int i, int n);
void write_picture(std::vector<uint8_t> &out, std::vector<size_t> *offs, const SliceParams &p, int slices, const x264gpu_mb *mbs, const int16_t *levels,
                   bool annexb, bool long_startcode_first, SliceStats *stats, int threads = 1, const x264gpu_level_index *index = nullptr);
// cabac.cpp: the same slice with CABAC slice data (write_slice dispatches on p.cabac)
void write_slice_cabac(std::vector<uint8_t> &out, const SliceParams &p, const x264gpu_mb *mbs, const int16_t *levels,
                       bool annexb, bool long_startcode, SliceStats *stats, const x264gpu_level_index *index = nullptr);

// ---- file output (muxers.cpp): the reference's cli_output_t (output/output.h: open_file / set_param / write_headers / write_frame / close_file) ----
class Muxer {
public:
    virtual ~Muxer() {}
    virtual int set_param(const x264_param_t *p) = 0;
    virtual int write_headers(const x264_nal_t *nal) = 0;                       // nal[0..2] = SPS, PPS, SEI of x264_encoder_headers
    virtual int write_frame(const uint8_t *payload, int size, const x264_picture_t *pic) = 0;
    virtual int close(int64_t largest_pts, int64_t second_largest_pts) = 0;
};
// muxer: "auto" (by extension), "raw", "mkv", "flv"; *annexb_out = how the encoder must frame NAL units for it.  NULL + *error on failure.
Muxer *open_muxer(const char *filename, const char *muxer, int *annexb_out, const char **error);

}  // namespace x264host

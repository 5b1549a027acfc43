// sets.cpp — sequence / picture parameter sets and the version SEI (ITU-T H.264 7.3.2.1, 7.3.2.2, E.1.1,
// D.1.6).  Plays the role of [x264-upstream] encoder/set.c behind x264_encoder_headers (codec.c:1650):
// nal[0]=SPS, nal[1]=PPS, nal[2]=SEI, each with a 4-byte prefix (output/raw.c:41-47 relies on it).
#include "host.hpp"
#include <string.h>

namespace x264host {

void write_sps(std::vector<uint8_t> &out, const SpsParams &s, bool annexb)
{
    BitWriter bw;
    bw.put((uint32_t)s.profile_idc, 8);
    bw.put1(s.constraint_set0); bw.put1(s.constraint_set1); bw.put(0, 6);      // constraint_set2..5 + reserved
    bw.put((uint32_t)s.level_idc, 8);
    bw.ue(s.sps_id);
    if (s.profile_idc >= 100) {                 // High: 4:2:0, 8 bit, no transform bypass, flat scaling matrices
        bw.ue(1);                               // chroma_format_idc
        bw.ue(0); bw.ue(0);                     // bit_depth_luma_minus8, bit_depth_chroma_minus8
        bw.put1(0);                             // qpprime_y_zero_transform_bypass_flag
        bw.put1(0);                             // seq_scaling_matrix_present_flag
    }
    bw.ue(s.log2_max_frame_num - 4);
    if (s.log2_max_poc_lsb > 0) { bw.ue(0); bw.ue(s.log2_max_poc_lsb - 4); }      // pic_order_cnt_type 0 (B pictures): log2_max_pic_order_cnt_lsb_minus4
    else bw.ue(2);                              // pic_order_cnt_type 2: output order == decoding order (no B frames)
    bw.ue(s.num_ref_frames);
    bw.put1(0);                                 // gaps_in_frame_num_value_allowed_flag
    bw.ue(s.mbw - 1);
    bw.ue(s.mbh - 1);
    bw.put1(1);                                 // frame_mbs_only_flag
    bw.put1(1);                                 // direct_8x8_inference_flag
    bool crop = s.crop_right || s.crop_bottom;
    bw.put1(crop);
    if (crop) { bw.ue(0); bw.ue(s.crop_right / 2); bw.ue(0); bw.ue(s.crop_bottom / 2); }
    bw.put1(1);                                 // vui_parameters_present_flag
    {
        bool sar = s.sar_w > 0 && s.sar_h > 0;
        bw.put1(sar);
        if (sar) {
            static const uint8_t tab[][2] = { { 1, 1 }, { 12, 11 }, { 10, 11 }, { 16, 11 }, { 40, 33 }, { 24, 11 }, { 20, 11 }, { 32, 11 },
                                              { 80, 33 }, { 18, 11 }, { 15, 11 }, { 64, 33 }, { 160, 99 }, { 4, 3 }, { 3, 2 }, { 2, 1 } };
            int idc = 255;
            for (int i = 0; i < 16; i++) if (tab[i][0] == s.sar_w && tab[i][1] == s.sar_h) idc = i + 1;
            bw.put((uint32_t)idc, 8);
            if (idc == 255) { bw.put((uint32_t)s.sar_w, 16); bw.put((uint32_t)s.sar_h, 16); }
        }
        bw.put1(s.overscan > 0);
        if (s.overscan > 0) bw.put1(s.overscan == 2);
        bool colour = (s.colorprim >= 0 && s.colorprim != 2) || (s.transfer >= 0 && s.transfer != 2) || (s.colmatrix >= 0 && s.colmatrix != 2);
        bool signal = s.fullrange > 0 || (s.vidformat >= 0 && s.vidformat != 5) || colour;
        bw.put1(signal);
        if (signal) {
            bw.put((uint32_t)(s.vidformat >= 0 ? s.vidformat : 5), 3);
            bw.put1(s.fullrange > 0);
            bw.put1(colour);
            if (colour) {
                bw.put((uint32_t)(s.colorprim >= 0 ? s.colorprim : 2), 8);
                bw.put((uint32_t)(s.transfer >= 0 ? s.transfer : 2), 8);
                bw.put((uint32_t)(s.colmatrix >= 0 ? s.colmatrix : 2), 8);
            }
        }
        bw.put1(0);                             // chroma_loc_info_present_flag
        bool timing = s.num_units_in_tick > 0 && s.time_scale > 0;
        bw.put1(timing);
        if (timing) { bw.put(s.num_units_in_tick, 32); bw.put(s.time_scale, 32); bw.put1(1); }   // fixed_frame_rate_flag
        bw.put1(0); bw.put1(0);                 // nal_hrd, vcl_hrd
        bw.put1(0);                             // pic_struct_present_flag
        bw.put1(1);                             // bitstream_restriction_flag
        bw.put1(1);                             // motion_vectors_over_pic_boundaries_flag
        bw.ue(0); bw.ue(0);                     // max_bytes_per_pic_denom, max_bits_per_mb_denom
        { int v = 4 * (s.mv_range > 0 ? s.mv_range : 512) - 1, l = 0; while (v >> (l + 1)) l++;      // x264: (int)log2f(max(1, mv_range * 4 - 1)) + 1
          bw.ue((uint32_t)(l + 1)); bw.ue((uint32_t)(l + 1)); }   // log2_max_mv_length_horizontal / vertical
        bw.ue(s.num_reorder_frames);            // max_num_reorder_frames
        bw.ue(s.num_ref_frames);                // max_dec_frame_buffering
    }
    bw.trailing();
    append_nal(out, 3, 7, bw.bytes(), annexb, true);
}

void write_pps(std::vector<uint8_t> &out, const PpsParams &p, bool annexb)
{
    BitWriter bw;
    bw.ue(p.pps_id);
    bw.ue(p.sps_id);
    bw.put1(p.cabac);
    bw.put1(0);                                 // bottom_field_pic_order_in_frame_present_flag
    bw.ue(0);                                   // num_slice_groups_minus1
    bw.ue(p.num_ref - 1);                       // num_ref_idx_l0_default_active_minus1
    bw.ue(0);                                   // l1
    bw.put1(p.weighted_pred);                   // weighted_pred_flag (x264: --weightp > 0)
    bw.put((uint32_t)p.weighted_bipred_idc, 2); // weighted_bipred_idc (2: implicit, x264 --weightb)
    bw.se(p.pic_init_qp - 26);
    bw.se(0);                                   // pic_init_qs_minus26
    bw.se(p.chroma_qp_offset);
    bw.put1(1);                                 // deblocking_filter_control_present_flag
    bw.put1(0);                                 // constrained_intra_pred_flag
    bw.put1(0);                                 // redundant_pic_cnt_present_flag
    if (p.transform8x8_mode) {
        bw.put1(1);                             // transform_8x8_mode_flag
        bw.put1(0);                             // pic_scaling_matrix_present_flag
        bw.se(p.chroma_qp_offset);              // second_chroma_qp_index_offset
    }
    bw.trailing();
    append_nal(out, 3, 8, bw.bytes(), annexb, true);
}

void write_sei_version(std::vector<uint8_t> &out, const char *text, bool annexb)
{
    // user_data_unregistered (payloadType 5): 16-byte UUID + text
    static const uint8_t uuid[16] = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x78, 0x32, 0x36, 0x34, 0x76, 0x66, 0x77, 0x2d, 0x31 };
    BitWriter bw;
    size_t len = 16 + strlen(text) + 1;
    bw.put(5, 8);
    size_t l = len;
    while (l >= 255) { bw.put(255, 8); l -= 255; }
    bw.put((uint32_t)l, 8);
    for (int i = 0; i < 16; i++) bw.put(uuid[i], 8);
    for (const char *c = text; *c; c++) bw.put((uint8_t)*c, 8);
    bw.put(0, 8);
    bw.trailing();
    append_nal(out, 0, 6, bw.bytes(), annexb, true);
}

}  // namespace x264host

/* oracle/lsmash_shim.c — TEST INFRASTRUCTURE ONLY.
 *
 * Thin driver around the H.264 parameter-set / slice-header parser of L-SMASH, third-party code that ships inside the
 * reference tree (/root/reference/output/L-SMASH, used by the reference's mp4 muxer output/mp4_lsmash.c:311-372).  The recipe
 * in oracle/Makefile compiles L-SMASH's own sources WHERE THEY LIE together with this file into oracle/_ref/liblsmash_ref.so
 * (nothing of L-SMASH is copied into this repository; the library is git-ignored and only exists where /root/reference is
 * mounted or where a prebuilt copy travelled to).  It gives the bitstream headers this encoder emits an independent reader:
 * what x264_encoder_headers / the slice writer meant must be what L-SMASH parses (tests/test_lsmash_ref.py).
 * It does not look at macroblock data — slice_data() parity stays with oracle/h264dec.cpp.
 * x264o_lsmash_read_mp4 is the same library as DEMUXER (its public reading API, lsmash.h): the mp4 files of the product's own
 * muxer (x264vfw_amd/host/muxers.cpp) must read back, through the code the reference's mp4 output is built on, as the samples,
 * timestamps, sync flags and parameter sets that went in (tests/test_muxers_cpu.py).
 */
#include "common/internal.h" /* must be placed first (L-SMASH convention) */
#include <string.h>
#include <stdlib.h>
#include "core/box.h"
#include "codecs/h264.h"
#include "codecs/nalu.h"

typedef struct {
    int32_t profile_idc, constraint_set_flags, level_idc, sps_id, chroma_format_idc, log2_max_frame_num, pic_order_cnt_type,
            max_num_ref_frames, frame_mbs_only_flag, cropped_width, cropped_height, sar_width, sar_height, video_full_range_flag,
            colour_primaries, transfer_characteristics, matrix_coefficients, fixed_frame_rate_flag;
    uint32_t num_units_in_tick, time_scale;
} x264o_ls_sps;
typedef struct {
    int32_t pps_id, sps_id, entropy_coding_mode_flag, num_ref_idx_l0_default_active_minus1, weighted_pred_flag, weighted_bipred_idc,
            deblocking_filter_control_present_flag, redundant_pic_cnt_present_flag;
} x264o_ls_pps;
typedef struct { int32_t nal_unit_type, nal_ref_idc, slice_type, idr, pps_id, frame_num, idr_pic_id; } x264o_ls_slice;

/* Annex-B stream -> last SPS / PPS seen and up to max_slices slice headers.  Returns the number of NAL units parsed, < 0 on a
 * parse error (the L-SMASH error code). */
int x264o_lsmash_parse_annexb(const uint8_t *buf, size_t n, x264o_ls_sps *osps, x264o_ls_pps *opps, x264o_ls_slice *oslices,
                              int max_slices, int *nslices)
{
    h264_info_t info;
    int err = h264_setup_parser(&info, 1), nals = 0, ns = 0;
    if (err < 0) return err;
    size_t i = 0;
    while (i + 3 < n) {
        if (!(buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1)) { i++; continue; }
        size_t start = i + 3, end = start;
        while (end + 2 < n && !(buf[end] == 0 && buf[end + 1] == 0 && (buf[end + 2] == 1 || (buf[end + 2] == 0 && end + 3 < n && buf[end + 3] == 1)))) end++;
        if (end + 2 >= n) end = n;
        if (end > start) {
            h264_nalu_header_t nuh;
            nuh.forbidden_zero_bit = buf[start] >> 7; nuh.nal_ref_idc = (buf[start] >> 5) & 3; nuh.nal_unit_type = buf[start] & 31; nuh.length = 1;
            uint8_t *ebsp = (uint8_t *)buf + start + 1;
            uint64_t sz = end - start - 1;
            err = 0;
            if (nuh.nal_unit_type == H264_NALU_TYPE_SPS) err = h264_parse_sps(&info, info.buffer.rbsp, ebsp, sz);
            else if (nuh.nal_unit_type == H264_NALU_TYPE_PPS) err = h264_parse_pps(&info, info.buffer.rbsp, ebsp, sz);
            else if (nuh.nal_unit_type == H264_NALU_TYPE_SLICE_N_IDR || nuh.nal_unit_type == H264_NALU_TYPE_SLICE_IDR) {
                err = h264_parse_slice(&info, &nuh, info.buffer.rbsp, ebsp, sz);
                if (err >= 0 && ns < max_slices) {
                    x264o_ls_slice *s = &oslices[ns++];
                    s->nal_unit_type = nuh.nal_unit_type; s->nal_ref_idc = nuh.nal_ref_idc; s->slice_type = info.slice.type;
                    s->idr = info.slice.IdrPicFlag; s->pps_id = info.slice.pic_parameter_set_id; s->frame_num = (int32_t)info.slice.frame_num;
                    s->idr_pic_id = info.slice.idr_pic_id;
                }
            }
            if (err < 0) { h264_cleanup_parser(&info); return err; }
            nals++;
        }
        i = end;
    }
    const h264_sps_t *sps = &info.sps;      /* active sets: copied by the slice parser */
    const h264_pps_t *pps = &info.pps;
    osps->profile_idc = sps->profile_idc; osps->constraint_set_flags = sps->constraint_set_flags; osps->level_idc = sps->level_idc;
    osps->sps_id = sps->seq_parameter_set_id; osps->chroma_format_idc = sps->chroma_format_idc; osps->log2_max_frame_num = sps->log2_max_frame_num;
    osps->pic_order_cnt_type = sps->pic_order_cnt_type; osps->max_num_ref_frames = (int32_t)sps->max_num_ref_frames;
    osps->frame_mbs_only_flag = sps->frame_mbs_only_flag; osps->cropped_width = (int32_t)sps->cropped_width; osps->cropped_height = (int32_t)sps->cropped_height;
    osps->sar_width = sps->vui.sar_width; osps->sar_height = sps->vui.sar_height; osps->video_full_range_flag = sps->vui.video_full_range_flag;
    osps->colour_primaries = sps->vui.colour_primaries; osps->transfer_characteristics = sps->vui.transfer_characteristics;
    osps->matrix_coefficients = sps->vui.matrix_coefficients; osps->fixed_frame_rate_flag = sps->vui.fixed_frame_rate_flag;
    osps->num_units_in_tick = sps->vui.num_units_in_tick; osps->time_scale = sps->vui.time_scale;
    opps->pps_id = pps->pic_parameter_set_id; opps->sps_id = pps->seq_parameter_set_id; opps->entropy_coding_mode_flag = pps->entropy_coding_mode_flag;
    opps->num_ref_idx_l0_default_active_minus1 = pps->num_ref_idx_l0_default_active_minus1; opps->weighted_pred_flag = pps->weighted_pred_flag;
    opps->weighted_bipred_idc = pps->weighted_bipred_idc; opps->deblocking_filter_control_present_flag = pps->deblocking_filter_control_present_flag;
    opps->redundant_pic_cnt_present_flag = pps->redundant_pic_cnt_present_flag;
    *nslices = ns;
    h264_cleanup_parser(&info);
    return nals;
}


/* ---- mp4 read-back through L-SMASH's demuxing API ---- */
typedef struct {
    uint32_t movie_timescale, media_timescale, n_samples, width, height, par_h, par_v, n_edits, avcc_size;
    uint64_t movie_duration, media_duration, track_duration, edit_duration;
    int64_t edit_start_time;
    int32_t edit_rate, primaries, transfer, matrix, full_range;
    uint32_t display_width, display_height;
    uint8_t avcc[512];                         /* the avcC box L-SMASH rebuilds from what it parsed (unstructured form) */
} x264o_mp4_info;
typedef struct { uint64_t dts, cts, pos; uint32_t length, sync; } x264o_mp4_sample;

/* Returns the number of samples (<= max_samples described; data of all of them concatenated into data_out up to data_cap), < 0 on error */
int x264o_lsmash_read_mp4(const char *path, x264o_mp4_info *info, x264o_mp4_sample *samples, int max_samples, uint8_t *data_out, size_t data_cap)
{
    memset(info, 0, sizeof(*info));
    lsmash_root_t *root = lsmash_create_root();
    if (!root) return -1;
    lsmash_file_parameters_t fp;
    if (lsmash_open_file(path, 1, &fp) < 0) { lsmash_destroy_root(root); return -2; }
    int ret = -3;
    lsmash_file_t *file = lsmash_set_file(root, &fp);
    if (!file || lsmash_read_file(file, &fp) < 0) goto done;
    lsmash_movie_parameters_t mv;
    lsmash_initialize_movie_parameters(&mv);
    if (lsmash_get_movie_parameters(root, &mv) < 0) { ret = -4; goto done; }
    info->movie_timescale = mv.timescale; info->movie_duration = mv.duration;
    uint32_t track = lsmash_get_track_ID(root, 1);
    if (!track) { ret = -5; goto done; }
    lsmash_track_parameters_t tp;
    lsmash_initialize_track_parameters(&tp);
    if (lsmash_get_track_parameters(root, track, &tp) < 0) { ret = -6; goto done; }
    info->track_duration = tp.duration; info->display_width = tp.display_width; info->display_height = tp.display_height;
    lsmash_media_parameters_t md;
    lsmash_initialize_media_parameters(&md);
    if (lsmash_get_media_parameters(root, track, &md) < 0) { ret = -7; goto done; }
    info->media_timescale = md.timescale; info->media_duration = md.duration;
    info->n_edits = lsmash_count_explicit_timeline_map(root, track);
    if (info->n_edits) {
        lsmash_edit_t e;
        if (lsmash_get_explicit_timeline_map(root, track, 1, &e) < 0) { ret = -8; goto done; }
        info->edit_duration = e.duration; info->edit_start_time = e.start_time; info->edit_rate = e.rate;
    }
    lsmash_summary_t *sum = lsmash_get_summary(root, track, 1);
    if (!sum) { ret = -9; goto done; }
    if (sum->summary_type == LSMASH_SUMMARY_TYPE_VIDEO) {
        lsmash_video_summary_t *v = (lsmash_video_summary_t *)sum;
        info->width = v->width; info->height = v->height; info->par_h = v->par_h; info->par_v = v->par_v;
        info->primaries = v->color.primaries_index; info->transfer = v->color.transfer_index; info->matrix = v->color.matrix_index;
        info->full_range = v->color.full_range;
    }
    for (uint32_t i = 1; i <= lsmash_count_codec_specific_data(sum); i++) {
        lsmash_codec_specific_t *cs = lsmash_get_codec_specific_data(sum, i);
        if (!cs || cs->type != LSMASH_CODEC_SPECIFIC_DATA_TYPE_ISOM_VIDEO_H264) continue;
        lsmash_codec_specific_t *u = cs->format == LSMASH_CODEC_SPECIFIC_FORMAT_UNSTRUCTURED ? cs : lsmash_convert_codec_specific_format(cs, LSMASH_CODEC_SPECIFIC_FORMAT_UNSTRUCTURED);
        if (u && u->size <= sizeof(info->avcc)) { info->avcc_size = u->size; memcpy(info->avcc, u->data.unstructured, u->size); }
        if (u && u != cs) lsmash_destroy_codec_specific_data(u);
    }
    lsmash_cleanup_summary(sum);
    if (lsmash_construct_timeline(root, track) < 0) { ret = -10; goto done; }
    info->n_samples = lsmash_get_sample_count_in_media_timeline(root, track);
    size_t used = 0;
    for (uint32_t i = 1; i <= info->n_samples; i++) {
        lsmash_sample_t *sm = lsmash_get_sample_from_media_timeline(root, track, i);
        if (!sm) { ret = -11; goto done; }
        if ((int)i <= max_samples) {
            x264o_mp4_sample *o = &samples[i - 1];
            o->dts = sm->dts; o->cts = sm->cts; o->pos = sm->pos; o->length = sm->length;
            o->sync = (sm->prop.ra_flags & ISOM_SAMPLE_RANDOM_ACCESS_FLAG_SYNC) != 0;
        }
        if (used + sm->length <= data_cap) { memcpy(data_out + used, sm->data, sm->length); used += sm->length; }
        lsmash_delete_sample(sm);
    }
    lsmash_destruct_timeline(root, track);
    ret = (int)info->n_samples;
done:
    lsmash_close_file(&fp);
    lsmash_destroy_root(root);
    return ret;
}

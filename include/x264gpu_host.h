/* x264gpu_host.h — symbols libx264gpu_host.so exports BESIDE the x264 API of include/x264.h and DriverProc of
 * include/vfw_shim.h.  One product entry point (the zero-copy input used by the VfW shell after the device-side
 * colourspace conversion) and the diagnostic hooks the parity tests use to drive the host entropy coder with
 * caller-supplied macroblock records.  C ABI, plain pointers and sizes. */
#ifndef X264GPU_HOST_H
#define X264GPU_HOST_H
#include <stdint.h>
#include "x264.h"
#include "x264gpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Device pointer of the encoder's input staging buffer: one tight I420 picture (Y w*h, then U, then V).
 * x264_encoder_encode() given a picture whose img.plane[0] equals this pointer encodes it in place — no host copy-in,
 * no upload.  Replaces the conv_pic hand-off of codec.c:1774-1786 when the conversion ran on the device. */
uint8_t *x264gpu_host_input_i420(x264_t *h);

/* ---- diagnostics (tests only) ---- */
/* entropy-code one slice from caller-supplied records/levels; returns bytes written (Annex-B NAL) or -1 */
int x264host_write_slice(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                         int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode,
                         const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped);
/* SPS + PPS for a stream made of such slices */
int x264host_write_headers(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset,
                           uint32_t num_units_in_tick, uint32_t time_scale, int num_ref, int transform8x8_mode, uint8_t *out, int cap);
/* the same pair with CABAC slice data (PPS entropy_coding_mode_flag = 1; Main / High profile).  slices = N: N slices as x264's slice threads code
 * them (disable_deblocking_filter_idc 2 where the caller passes 0); slices = -N: N slices as --slices N codes them (idc stays 0) */
int x264host_write_picture(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                           int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode, int cabac, int slices,
                           const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped);
int x264host_write_slice_cabac(int mbw, int mbh, int slice_type, int qp, int pic_init_qp, int frame_num, int log2_max_frame_num,
                               int idr, int idr_pic_id, int disable_deblock_idc, int num_ref, int num_ref_default, int transform8x8_mode,
                               const x264gpu_mb *mbs, const int16_t *levels, uint8_t *out, int cap, int *skipped);
int x264host_write_headers_cabac(int width, int height, int level_idc, int log2_max_frame_num, int pic_init_qp, int chroma_qp_offset,
                                 uint32_t num_units_in_tick, uint32_t time_scale, int num_ref, int transform8x8_mode, int cabac, uint8_t *out, int cap);
/* (pStateIdx << 1) | valMPS of the 460 context variables after the last CABAC slice the calling thread wrote */
void x264host_cabac_last_states(uint8_t *out460);
/* the file muxers (`--output x.h264 | x.mkv | x.flv`, host/muxers.cpp; reference output/raw.c, matroska.c, flv.c) without an encoder */
void *x264host_mux_open(const char *filename, const char *muxer /* "auto", "raw", "mkv", "flv" */, int *annexb);
int x264host_mux_set_param(void *h, int width, int height, uint32_t fps_num, uint32_t fps_den, uint32_t timebase_num, uint32_t timebase_den,
                           int sar_width, int sar_height, int vfr);
int x264host_mux_write_headers(void *h, const uint8_t *sps, int sps_size, const uint8_t *pps, int pps_size, const uint8_t *sei, int sei_size);
int x264host_mux_write_frame(void *h, const uint8_t *payload, int size, int64_t pts, int64_t dts, int keyframe, int type);
int x264host_mux_close(void *h, int64_t largest_pts, int64_t second_largest_pts);
/* quantiser, scenecut flag and lookahead sums (x264gpu_lookahead_frame_cost) of the last coded picture */
int x264host_last_decision(x264_t *h, int *qp, int *scenecut, int32_t costs[4]);
/* ... and the float quantiser (x264 rc->qpm) its macroblock quantisers were rounded from (0 = the integer quantiser) */
float x264host_last_qpm(x264_t *h);
/* ... the second pass' plan (init_pass2): per picture of the statistics file, display order, the planned quantiser scale and the bits expected before it; returns the count */
int x264host_pass2_plan(x264_t *h, double *new_qscale, double *expected_bits, int n);
/* how many pictures of the session may be in flight on the device at once (launch contexts over the shared DPB; 1: one picture a call) */
int x264host_pictures_in_flight(x264_t *h);
/* reconstructed picture of the last encoded frame as I420 (host memory) */
int x264host_get_recon(x264_t *h, uint8_t *i420_out);

#ifdef __cplusplus
}
#endif
#endif

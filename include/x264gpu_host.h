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
/* quantiser, scenecut flag and lookahead sums (x264gpu_lookahead_frame_cost) of the last coded picture */
int x264host_last_decision(x264_t *h, int *qp, int *scenecut, int32_t costs[4]);
/* reconstructed picture of the last encoded frame as I420 (host memory) */
int x264host_get_recon(x264_t *h, uint8_t *i420_out);

#ifdef __cplusplus
}
#endif
#endif

// stubs.hip — entry points declared in include/x264gpu.h that are not implemented yet fail loudly.
#include "common.cuh"
using namespace x264gpu;
#define NOTIMPL(name) return set_err(X264GPU_EINVAL, name ": not implemented", hipSuccess)
extern "C" {
int x264gpu_dctq8x8(const uint8_t *, const uint8_t *, int, int, int, int16_t *, int16_t *, uint8_t *, void *) { NOTIMPL("x264gpu_dctq8x8"); }
int x264gpu_intra_predict(int, const uint8_t *, int, const int32_t *, const int32_t *, const int32_t *, int, uint8_t *, void *) { NOTIMPL("x264gpu_intra_predict"); }
int x264gpu_encoder_create(x264gpu_encoder **, const x264gpu_config *) { NOTIMPL("x264gpu_encoder_create"); }
void x264gpu_encoder_destroy(x264gpu_encoder *) {}
int x264gpu_encoder_mb_count(const x264gpu_encoder *) { return 0; }
int x264gpu_encode_frames(x264gpu_encoder *, const uint8_t *, int, x264gpu_mb *, int16_t *, void *) { NOTIMPL("x264gpu_encode_frames"); }
int x264gpu_encoder_get_recon(x264gpu_encoder *, int, uint8_t *, void *) { NOTIMPL("x264gpu_encoder_get_recon"); }
int x264gpu_encoder_stage_count(void) { return 0; }
const char *x264gpu_encoder_stage_name(int) { return ""; }
}

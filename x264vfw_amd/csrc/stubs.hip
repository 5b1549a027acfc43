// stubs.hip — entry points declared in include/x264gpu.h that are not implemented yet fail loudly.
#include "common.cuh"
using namespace x264gpu;
#define NOTIMPL(name) return set_err(X264GPU_EINVAL, name ": not implemented", hipSuccess)
extern "C" {
int x264gpu_dctq8x8(const uint8_t *, const uint8_t *, int, int, int, int16_t *, int16_t *, uint8_t *, void *) { NOTIMPL("x264gpu_dctq8x8"); }
int x264gpu_intra_predict(int, const uint8_t *, int, const int32_t *, const int32_t *, const int32_t *, int, uint8_t *, void *) { NOTIMPL("x264gpu_intra_predict"); }
}

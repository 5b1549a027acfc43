// stubs.hip — entry points declared in include/x264gpu.h that are not implemented yet fail loudly.
#include "common.cuh"
using namespace x264gpu;
extern "C" {
int x264gpu_dctq8x8(const uint8_t *, const uint8_t *, int, int, int, int16_t *, int16_t *, uint8_t *, void *)
{ return set_err(X264GPU_EINVAL, "x264gpu_dctq8x8: not implemented", hipSuccess); }
}

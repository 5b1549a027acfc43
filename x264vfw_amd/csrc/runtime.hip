// runtime.hip — C-ABI runtime shims (device selection, memory, streams) of include/x264gpu.h.
#include "common.hip.h"
#include <string.h>

namespace x264gpu {
thread_local char g_err[256] = "";
int set_err(int code, const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    return code;
}
}  // namespace x264gpu
using namespace x264gpu;

extern "C" {

int x264gpu_abi_version(void) { return X264GPU_ABI_VERSION; }

int x264gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int x264gpu_set_device(int dev) { HIP_TRY(hipSetDevice(dev)); return X264GPU_OK; }
int x264gpu_get_device(int *dev) { ARG_TRY(dev); HIP_TRY(hipGetDevice(dev)); return X264GPU_OK; }

const char *x264gpu_last_error(void) { return g_err; }

int x264gpu_malloc(void **d_ptr, size_t bytes)
{
    if (!d_ptr) return set_err(X264GPU_EINVAL, "x264gpu_malloc", hipSuccess);
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 1);
    if (e != hipSuccess) return set_err(e == hipErrorOutOfMemory ? X264GPU_ENOMEM : X264GPU_EHIP, "hipMalloc", e);
    return X264GPU_OK;
}
int x264gpu_free(void *d_ptr) { HIP_TRY(hipFree(d_ptr)); return X264GPU_OK; }
int x264gpu_memcpy_h2d(void *d, const void *h, size_t n, void *stream)
{
    HIP_TRY(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return X264GPU_OK;
}
int x264gpu_memcpy_d2h(void *h, const void *d, size_t n, void *stream)
{
    HIP_TRY(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return X264GPU_OK;
}
int x264gpu_stream_create(void **stream)
{
    ARG_TRY(stream);
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return X264GPU_OK;
}
int x264gpu_stream_destroy(void *stream)
{
    ARG_TRY(stream);
    HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return X264GPU_OK;
}
int x264gpu_event_create(void **event)
{
    ARG_TRY(event);
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventBlockingSync));      // a waiting host thread sleeps (the callers need the cores)
    *event = (void *)ev;
    return X264GPU_OK;
}
int x264gpu_event_destroy(void *event) { ARG_TRY(event); HIP_TRY(hipEventDestroy((hipEvent_t)event)); return X264GPU_OK; }
int x264gpu_event_record(void *event, void *stream) { ARG_TRY(event); HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return X264GPU_OK; }
int x264gpu_stream_wait_event(void *stream, void *event) { ARG_TRY(event); HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0)); return X264GPU_OK; }
int x264gpu_event_sync(void *event) { ARG_TRY(event); HIP_TRY(hipEventSynchronize((hipEvent_t)event)); return X264GPU_OK; }
int x264gpu_memcpy_d2d(void *dst, const void *src, size_t n, void *stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return X264GPU_OK;
}
int x264gpu_memset(void *d, int v, size_t n, void *stream)
{
    HIP_TRY(hipMemsetAsync(d, v, n, (hipStream_t)stream));
    return X264GPU_OK;
}
int x264gpu_stream_sync(void *stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return X264GPU_OK; }

}  // extern "C"

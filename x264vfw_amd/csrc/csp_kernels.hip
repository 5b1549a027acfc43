// csp_kernels.hip — input colourspace conversion to I420 on the device (replaces the
// x264vfw_csp_function_t table of /root/reference/csp.c:436-487 for an I420 encoder, codec.c:1799 call site).
// Pure streaming kernels, HBM-bound: every source byte is read once with dword-or-wider loads, every output
// byte written once.  A thread owns an 8-pixel x 2-row cell (one 4:2:0 chroma row), so luma leaves as 8-byte
// stores and chroma as 4-byte stores; ragged right edges (width % 8) take a scalar tail.  The vertical flip is a
// negative source pitch, exactly as in the reference.  Restates oracle/csp.c bit-exactly.
#include "common.hip.h"

using namespace x264gpu;

namespace {

struct CspArgs {
    const uint8_t *src[3]; long sstride[3];      // source planes (already flip-adjusted: first row + signed pitch)
    uint8_t *dst[3]; int dstride[3];
    int w, h;                                    // luma size (even)
    long sframe, dframe;                         // byte distance between consecutive frames of a batch (blockIdx.z)
    uint32_t c[12];                              // RGB coefficients (oracle x264o_csp_rgb_coefs)
};

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void st32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
__device__ __forceinline__ void st64(uint8_t *p, uint32_t a, uint32_t b) { uint2 v = make_uint2(a, b); __builtin_memcpy(p, &v, 8); }

// rounding average of 4 packed bytes: (a+b+1)>>1
__device__ __forceinline__ uint32_t avg4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }

// ---- plane copy / 2:1 vertical / 2:1 both (csp.c:28-74): thread = 16 output bytes of one row ----
template <int MODE>
__global__ __launch_bounds__(256) void k_csp_plane(uint8_t *__restrict__ dst, int ds, const uint8_t *__restrict__ src, long ss, int w, int h,
                                                   long sframe, long dframe)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 16, y = blockIdx.y;
    if (x >= w) return;
    dst += (long)blockIdx.z * dframe; src += (long)blockIdx.z * sframe;
    uint8_t *d = dst + (long)y * ds + x;
    const uint8_t *s = src + (long)y * (MODE == 0 ? 1 : 2) * ss + (MODE == 2 ? 2 * x : x);
    if (x + 16 <= w) {
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (MODE == 0) o[i] = ld32(s + 4 * i);
            else if (MODE == 1) o[i] = avg4(ld32(s + 4 * i), ld32(s + ss + 4 * i));
            else {
                const uint32_t a0 = ld32(s + 8 * i), a1 = ld32(s + 8 * i + 4), b0 = ld32(s + ss + 8 * i), b1 = ld32(s + ss + 8 * i + 4);
                uint32_t r = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t a = k < 2 ? a0 : a1, b = k < 2 ? b0 : b1;
                    const int sh = (k & 1) * 16;
                    r |= ((((a >> sh) & 0xff) + ((a >> (sh + 8)) & 0xff) + ((b >> sh) & 0xff) + ((b >> (sh + 8)) & 0xff) + 2) >> 2) << (8 * k);
                }
                o[i] = r;
            }
        }
        uint4 v = make_uint4(o[0], o[1], o[2], o[3]);
        __builtin_memcpy(d, &v, 16);
    } else {
        for (int i = 0; x + i < w; i++)
            d[i] = MODE == 0 ? s[i] : MODE == 1 ? (uint8_t)((s[i] + s[i + ss] + 1) >> 1)
                                                : (uint8_t)((s[2 * i] + s[2 * i + 1] + s[2 * i + ss] + s[2 * i + 1 + ss] + 2) >> 2);
    }
}

// ---- YUY2 / UYVY -> I420 (csp.c:155-205): thread = 8 pixels x 2 rows ----
template <bool UYVY>
__global__ __launch_bounds__(256) void k_csp_yuyv(CspArgs a)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y * 2;
    if (x >= a.w) return;
    const long fs = (long)blockIdx.z * a.sframe, fd = (long)blockIdx.z * a.dframe;
    const uint8_t *s0 = a.src[0] + fs + (long)r * a.sstride[0] + 2 * x, *s1 = s0 + a.sstride[0];
    uint8_t *y0 = a.dst[0] + fd + (long)r * a.dstride[0] + x, *y1 = y0 + a.dstride[0];
    uint8_t *u = a.dst[1] + fd + (long)(r >> 1) * a.dstride[1] + (x >> 1), *v = a.dst[2] + fd + (long)(r >> 1) * a.dstride[2] + (x >> 1);
    if (x + 8 <= a.w) {
        uint32_t p0[4], p1[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { p0[i] = ld32(s0 + 4 * i); p1[i] = ld32(s1 + 4 * i); }
        uint32_t ya[2] = { 0, 0 }, yb[2] = { 0, 0 }, uu = 0, vv = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {                                  // dword i = pixels 2i, 2i+1: Y0 U Y1 V  or  U Y0 V Y1
            const uint32_t q0 = UYVY ? p0[i] >> 8 : p0[i], q1 = UYVY ? p1[i] >> 8 : p1[i];
            ya[i >> 1] |= ((q0 & 0xff) | ((q0 >> 8) & 0xff00)) << (16 * (i & 1));
            yb[i >> 1] |= ((q1 & 0xff) | ((q1 >> 8) & 0xff00)) << (16 * (i & 1));
            const uint32_t c0 = UYVY ? p0[i] : p0[i] >> 8, c1 = UYVY ? p1[i] : p1[i] >> 8;   // byte0 = U, byte2 = V
            uu |= (((c0 & 0xff) + (c1 & 0xff) + 1) >> 1) << (8 * i);
            vv |= ((((c0 >> 16) & 0xff) + ((c1 >> 16) & 0xff) + 1) >> 1) << (8 * i);
        }
        st64(y0, ya[0], ya[1]); st64(y1, yb[0], yb[1]);
        st32(u, uu); st32(v, vv);
    } else {
        const int yo = UYVY ? 1 : 0, uo = UYVY ? 0 : 1;
        for (int i = 0; x + i < a.w; i += 2) {
            const uint8_t *q0 = s0 + 2 * i, *q1 = s1 + 2 * i;
            y0[i] = q0[yo]; y0[i + 1] = q0[yo + 2]; y1[i] = q1[yo]; y1[i + 1] = q1[yo + 2];
            u[i >> 1] = (uint8_t)((q0[uo] + q1[uo] + 1) >> 1);
            v[i >> 1] = (uint8_t)((q0[uo + 2] + q1[uo + 2] + 1) >> 1);
        }
    }
}

// ---- BGR / BGRA -> I420, 20-bit fixed point (csp.c:252-388): thread = 8 pixels x 2 rows ----
template <int STEP>
__global__ __launch_bounds__(256) void k_csp_bgr(CspArgs a)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 8, r = blockIdx.y * 2;
    if (x >= a.w) return;
    const long fs = (long)blockIdx.z * a.sframe, fd = (long)blockIdx.z * a.dframe;
    const uint8_t *s0 = a.src[0] + fs + (long)r * a.sstride[0] + (long)x * STEP, *s1 = s0 + a.sstride[0];
    uint8_t *y0 = a.dst[0] + fd + (long)r * a.dstride[0] + x, *y1 = y0 + a.dstride[0];
    uint8_t *u = a.dst[1] + fd + (long)(r >> 1) * a.dstride[1] + (x >> 1), *v = a.dst[2] + fd + (long)(r >> 1) * a.dstride[2] + (x >> 1);
    const int n = min(8, a.w - x);
    uint32_t w0[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, w1[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };      // up to 32 source bytes per row
    if (n == 8) {
#pragma unroll
        for (int i = 0; i < 2 * STEP; i++) { w0[i] = ld32(s0 + 4 * i); w1[i] = ld32(s1 + 4 * i); }
    } else {
        for (int i = 0; i < n * STEP; i++) { w0[i >> 2] |= (uint32_t)s0[i] << (8 * (i & 3)); w1[i >> 2] |= (uint32_t)s1[i] << (8 * (i & 3)); }
    }
    uint32_t ya[2] = { 0, 0 }, yb[2] = { 0, 0 }, uu = 0, vv = 0;
#pragma unroll
    for (int px = 0; px < 8; px += 2) {
        uint32_t cr = 0, cg = 0, cb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int p = px + (k >> 1), bo = p * STEP;                       // byte offset of the pixel inside the row cell
            const uint32_t *wr = (k & 1) ? w1 : w0;
            // three consecutive bytes B,G,R starting at byte bo (may straddle two dwords for STEP == 3)
            const uint32_t lo = wr[bo >> 2], hi = wr[(bo >> 2) + ((bo & 3) ? 1 : 0)];
            const uint32_t t = (bo & 3) ? __builtin_amdgcn_alignbyte(hi, lo, bo & 3) : lo;
            const uint32_t b = t & 0xff, g = (t >> 8) & 0xff, rr = (t >> 16) & 0xff;
            cr += rr; cg += g; cb += b;
            const uint32_t yv = (a.c[3] + a.c[0] * rr + a.c[1] * g + a.c[2] * b) >> 20;
            if (k & 1) yb[p >> 2] |= (yv & 0xff) << (8 * (p & 3)); else ya[p >> 2] |= (yv & 0xff) << (8 * (p & 3));
        }
        uu |= (((a.c[7] + a.c[6] * cb - a.c[4] * cr - a.c[5] * cg) >> 22) & 0xff) << (4 * px);
        vv |= (((a.c[11] + a.c[8] * cr - a.c[9] * cg - a.c[10] * cb) >> 22) & 0xff) << (4 * px);
    }
    if (n == 8) { st64(y0, ya[0], ya[1]); st64(y1, yb[0], yb[1]); st32(u, uu); st32(v, vv); }
    else
        for (int i = 0; i < n; i++) {
            y0[i] = (uint8_t)(ya[i >> 2] >> (8 * (i & 3))); y1[i] = (uint8_t)(yb[i >> 2] >> (8 * (i & 3)));
            if (!(i & 1)) { u[i >> 1] = (uint8_t)(uu >> (4 * i)); v[i >> 1] = (uint8_t)(vv >> (4 * i)); }
        }
}

void rgb_coefs(int colmatrix709, int fullrange, uint32_t c[12])
{
    // host doubles, evaluated in the reference's order (csp.c:205-251 FIX() macros)
    const double kb = colmatrix709 ? 0.0722 : 0.114, kr = colmatrix709 ? 0.2126 : 0.299;
    const double kg = 1.0 - kb - kr, sb = 1.0 - kb, sr = 1.0 - kr;
    const double ky = fullrange ? 1.0 : 1.0 * 219.0 / 255.0;
    const double ku = fullrange ? 0.5 / sb : (0.5 / sb) * 224.0 / 255.0, kv = fullrange ? 0.5 / sr : (0.5 / sr) * 224.0 / 255.0;
    const double ay = fullrange ? 0.0 : 16.0, fix = 1 << 20, rnd = 1 << 19;
    const int bu = fullrange ? -1 : 0;
    auto F = [&](double f) { return (uint32_t)(f * fix + 0.5); };
    c[0] = F(kr * ky); c[1] = F(kg * ky); c[2] = F(kb * ky); c[3] = (uint32_t)(ay * fix + rnd + 0.5);
    c[4] = F(kr * ku); c[5] = F(kg * ku); c[6] = F(sb * ku); c[7] = (uint32_t)((128.0 * fix + rnd) * 4 + bu + 0.5);
    c[8] = F(sr * kv); c[9] = F(kg * kv); c[10] = F(kb * kv); c[11] = (uint32_t)((128.0 * fix + rnd) * 4 + bu + 0.5);
}

template <int MODE>
void launch_plane(uint8_t *dst, int ds, const uint8_t *src, long ss, int w, int h, long sframe, long dframe, int frames, hipStream_t st)
{
    hipLaunchKernelGGL(k_csp_plane<MODE>, dim3((w + 4095) / 4096, h, frames), dim3(256), 0, st, dst, ds, src, ss, w, h, sframe, dframe);
}

}  // namespace

extern "C" {

long x264gpu_csp_img_fill(int csp, int width, int height, long off[3], int stride[3])
{
    if (!off || !stride) return -1;
    off[0] = off[1] = off[2] = 0; stride[0] = stride[1] = stride[2] = 0;
    switch (csp & X264GPU_CSP_MASK) {
    case X264GPU_CSP_I420: case X264GPU_CSP_YV12:
        height = (height + 1) & ~1; width = (width + 1) & ~1;
        stride[0] = width; stride[1] = stride[2] = width / 2;
        off[1] = (long)stride[0] * height; off[2] = off[1] + (long)stride[1] * height / 2;
        return off[2] + (long)stride[2] * height / 2;
    case X264GPU_CSP_YV16:
        width = (width + 1) & ~1;
        stride[0] = width; stride[1] = stride[2] = width / 2;
        off[1] = (long)stride[0] * height; off[2] = off[1] + (long)stride[1] * height;
        return off[2] + (long)stride[2] * height;
    case X264GPU_CSP_YV24:
        stride[0] = stride[1] = stride[2] = width;
        off[1] = (long)width * height; off[2] = 2 * off[1];
        return 3 * off[1];
    case X264GPU_CSP_YUYV: case X264GPU_CSP_UYVY: width = (width + 1) & ~1; stride[0] = 2 * width; return (long)stride[0] * height;
    case X264GPU_CSP_BGR: stride[0] = (3 * width + 3) & ~3; return (long)stride[0] * height;
    case X264GPU_CSP_BGRA: stride[0] = 4 * width; return (long)stride[0] * height;
    default: return -1;
    }
}

int x264gpu_csp_to_i420_batch(const uint8_t *const d_src[3], const int src_stride[3], size_t src_frame_bytes, int csp, int width, int height,
                              int colmatrix709, int fullrange, uint8_t *const d_dst[3], const int dst_stride[3], size_t dst_frame_bytes,
                              int frames, void *stream)
{
    ARG_TRY(frames >= 1 && frames <= 65535);
    const long sframe = (long)src_frame_bytes, dframe = (long)dst_frame_bytes;
    ARG_TRY(d_src && src_stride && d_dst && dst_stride && d_src[0] && d_dst[0] && d_dst[1] && d_dst[2]);
    ARG_TRY(width > 0 && height > 0 && !(width & 1) && !(height & 1));
    const int id = csp & X264GPU_CSP_MASK, flip = (csp & X264GPU_CSP_VFLIP) != 0;
    hipStream_t st = (hipStream_t)stream;
    const int w = width, h = height, cw = w >> 1, ch = h >> 1;
    switch (id) {
    case X264GPU_CSP_I420: case X264GPU_CSP_YV12: case X264GPU_CSP_YV16: case X264GPU_CSP_YV24: {
        ARG_TRY(d_src[1] && d_src[2]);
        const int swap = id != X264GPU_CSP_I420, vs = (id == X264GPU_CSP_YV16 || id == X264GPU_CSP_YV24) ? 2 : 1;
        launch_plane<0>(d_dst[0], dst_stride[0], flip ? d_src[0] + (long)(h - 1) * src_stride[0] : d_src[0], flip ? -(long)src_stride[0] : src_stride[0], w, h, sframe, dframe, frames, st);
        for (int p = 1; p <= 2; p++) {
            uint8_t *d = d_dst[swap ? 3 - p : p];
            const int ds = dst_stride[swap ? 3 - p : p];
            const long ss = flip ? -(long)src_stride[p] : src_stride[p];
            const uint8_t *s = flip ? d_src[p] + (long)(vs * ch - 1) * src_stride[p] : d_src[p];
            if (id == X264GPU_CSP_YV16) launch_plane<1>(d, ds, s, ss, cw, ch, sframe, dframe, frames, st);
            else if (id == X264GPU_CSP_YV24) launch_plane<2>(d, ds, s, ss, cw, ch, sframe, dframe, frames, st);
            else launch_plane<0>(d, ds, s, ss, cw, ch, sframe, dframe, frames, st);
        }
        break;
    }
    case X264GPU_CSP_YUYV: case X264GPU_CSP_UYVY: case X264GPU_CSP_BGR: case X264GPU_CSP_BGRA: {
        CspArgs a = {};
        a.src[0] = flip ? d_src[0] + (long)(h - 1) * src_stride[0] : d_src[0];
        a.sstride[0] = flip ? -(long)src_stride[0] : src_stride[0];
        for (int p = 0; p < 3; p++) { a.dst[p] = d_dst[p]; a.dstride[p] = dst_stride[p]; }
        a.w = w; a.h = h; a.sframe = sframe; a.dframe = dframe;
        rgb_coefs(colmatrix709, fullrange, a.c);
        const dim3 grid((w + 2047) / 2048, ch, frames), blk(256);
        if (id == X264GPU_CSP_YUYV) hipLaunchKernelGGL(k_csp_yuyv<false>, grid, blk, 0, st, a);
        else if (id == X264GPU_CSP_UYVY) hipLaunchKernelGGL(k_csp_yuyv<true>, grid, blk, 0, st, a);
        else if (id == X264GPU_CSP_BGR) hipLaunchKernelGGL(k_csp_bgr<3>, grid, blk, 0, st, a);
        else hipLaunchKernelGGL(k_csp_bgr<4>, grid, blk, 0, st, a);
        break;
    }
    default:
        return set_err(X264GPU_EINVAL, "unsupported input colourspace for an I420 encoder (csp.c:441-487)", hipSuccess);
    }
    HIP_TRY(hipGetLastError());
    return X264GPU_OK;
}

int x264gpu_csp_to_i420(const uint8_t *const d_src[3], const int src_stride[3], int csp, int width, int height, int colmatrix709,
                        int fullrange, uint8_t *const d_dst[3], const int dst_stride[3], void *stream)
{
    return x264gpu_csp_to_i420_batch(d_src, src_stride, 0, csp, width, height, colmatrix709, fullrange, d_dst, dst_stride, 0, 1, stream);
}

}  // extern "C"

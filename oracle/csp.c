/* oracle/csp.c — CPU restatement of the driver's input colourspace conversion TO I420 (test infrastructure only;
 * the product never links it).  Follows /root/reference/csp.c: plane copies / 2:1 subsampling (csp.c:28-94),
 * YV12/YV16/YV24 -> I420 (csp.c:101-131,409-414), YUY2/UYVY -> I420 (csp.c:155-205,421-422), BGR/BGRA -> I420 in
 * 20-bit fixed point for BT.601/709 x TV/PC range (csp.c:252-388,427-434), all with the optional vertical flip
 * (negative source stride).  Only the conversions x264vfw_csp_init installs for an I420 encoder (csp.c:441-487) are
 * restated; the reference cannot be compiled here (x264vfw.h needs windows.h / vfw.h / x264.h), so this file is
 * pinned by the closed-form checks in tests/test_oracle_spec.py (colour-bar values, identities), not by the reference. */
#include "x264o.h"
#include <string.h>

static void plane_copy(uint8_t *dst, int ds, const uint8_t *src, long ss, int w, int h)
{
    for (; h > 0; h--, dst += ds, src += ss) memcpy(dst, src, (size_t)w);
}
static void plane_subsamplev2(uint8_t *dst, int ds, const uint8_t *src, long ss, int w, int h)
{
    for (; h > 0; h--, dst += ds, src += 2 * ss)
        for (int i = 0; i < w; i++) dst[i] = (uint8_t)((src[i] + src[i + ss] + 1) >> 1);
}
static void plane_subsamplehv2(uint8_t *dst, int ds, const uint8_t *src, long ss, int w, int h)
{
    for (; h > 0; h--, dst += ds, src += 2 * ss)
        for (int i = 0; i < w; i++) dst[i] = (uint8_t)((src[2 * i] + src[2 * i + 1] + src[2 * i + ss] + src[2 * i + 1 + ss] + 2) >> 2);
}

#define BITS 20
#define INT_FIX (1 << BITS)
#define INT_ROUND (INT_FIX >> 1)
#define FIX(f) ((uint32_t)((f) * INT_FIX + 0.5))

void x264o_csp_rgb_coefs(int colmatrix709, int fullrange, uint32_t c[12])
{
    /* c = { Y_R, Y_G, Y_B, Y_ADD, U_R, U_G, U_B, U_ADD, V_R, V_G, V_B, V_ADD } (csp.c:205-251) */
    const double kb = colmatrix709 ? 0.0722 : 0.114, kr = colmatrix709 ? 0.2126 : 0.299;
    const double kg = 1.0 - kb - kr, sb = 1.0 - kb, sr = 1.0 - kr;
    const double ky = fullrange ? 1.0 : 1.0 * 219.0 / 255.0;
    const double ku = fullrange ? 0.5 / sb : (0.5 / sb) * 224.0 / 255.0, kv = fullrange ? 0.5 / sr : (0.5 / sr) * 224.0 / 255.0;
    const double ay = fullrange ? 0.0 : 16.0;
    const int bu = fullrange ? -1 : 0;
    c[0] = FIX(kr * ky); c[1] = FIX(kg * ky); c[2] = FIX(kb * ky); c[3] = (uint32_t)(ay * INT_FIX + INT_ROUND + 0.5);
    c[4] = FIX(kr * ku); c[5] = FIX(kg * ku); c[6] = FIX(sb * ku); c[7] = (uint32_t)((128.0 * INT_FIX + INT_ROUND) * 4 + bu + 0.5);
    c[8] = FIX(sr * kv); c[9] = FIX(kg * kv); c[10] = FIX(kb * kv); c[11] = (uint32_t)((128.0 * INT_FIX + INT_ROUND) * 4 + bu + 0.5);
}

/* x264vfw_img_fill (codec.c:304-379): plane offsets / strides of a frame held in one contiguous buffer; returns bytes */
long x264o_csp_img_fill(int csp, int width, int height, long off[3], int stride[3])
{
    off[0] = off[1] = off[2] = 0; stride[0] = stride[1] = stride[2] = 0;
    switch (csp & X264O_CSP_MASK) {
    case X264O_CSP_I420: case X264O_CSP_YV12:
        height = (height + 1) & ~1; width = (width + 1) & ~1;
        stride[0] = width; stride[1] = stride[2] = width / 2;
        off[1] = (long)stride[0] * height; off[2] = off[1] + (long)stride[1] * height / 2;
        return off[2] + (long)stride[2] * height / 2;
    case X264O_CSP_YV16:
        width = (width + 1) & ~1;
        stride[0] = width; stride[1] = stride[2] = width / 2;
        off[1] = (long)stride[0] * height; off[2] = off[1] + (long)stride[1] * height;
        return off[2] + (long)stride[2] * height;
    case X264O_CSP_YV24:
        stride[0] = stride[1] = stride[2] = width;
        off[1] = (long)width * height; off[2] = 2 * off[1];
        return 3 * off[1];
    case X264O_CSP_YUYV: case X264O_CSP_UYVY:
        width = (width + 1) & ~1; stride[0] = 2 * width; return (long)stride[0] * height;
    case X264O_CSP_BGR: stride[0] = (3 * width + 3) & ~3; return (long)stride[0] * height;
    case X264O_CSP_BGRA: stride[0] = 4 * width; return (long)stride[0] * height;
    default: return -1;
    }
}

int x264o_csp_to_i420(uint8_t *const dst[3], const int dstride[3], const uint8_t *const src[3], const int sstride[3],
                      int csp, int w, int h, int colmatrix709, int fullrange)
{
    const int flip = (csp & X264O_CSP_VFLIP) != 0, id = csp & X264O_CSP_MASK;
    if (w <= 0 || h <= 0 || (w & 1) || (h & 1)) return -1;
    switch (id) {
    case X264O_CSP_I420: case X264O_CSP_YV12: case X264O_CSP_YV16: case X264O_CSP_YV24: {
        /* YUV_TO_YUV( name, func, swap, 1, 1 ): chroma planes w/2 x h/2; every source here is Y,V,U ordered except I420 */
        const int swap = id != X264O_CSP_I420;
        const int vs = id == X264O_CSP_YV16 || id == X264O_CSP_YV24 ? 2 : 1;      /* source chroma rows per output row */
        const int cw = w >> 1, ch = h >> 1;
        plane_copy(dst[0], dstride[0], flip ? src[0] + (long)(h - 1) * sstride[0] : src[0], flip ? -(long)sstride[0] : sstride[0], w, h);
        for (int p = 1; p <= 2; p++) {
            uint8_t *d = dst[swap ? 3 - p : p];
            const int ds = dstride[swap ? 3 - p : p];
            const long ss = flip ? -(long)sstride[p] : sstride[p];
            const uint8_t *s = flip ? src[p] + (long)(vs * ch - 1) * sstride[p] : src[p];
            if (id == X264O_CSP_YV16) plane_subsamplev2(d, ds, s, ss, cw, ch);
            else if (id == X264O_CSP_YV24) plane_subsamplehv2(d, ds, s, ss, cw, ch);
            else plane_copy(d, ds, s, ss, cw, ch);
        }
        return 0;
    }
    case X264O_CSP_YUYV: case X264O_CSP_UYVY: {
        const int y1 = id == X264O_CSP_YUYV ? 0 : 1, y2 = y1 + 2, up = id == X264O_CSP_YUYV ? 1 : 0, vp = up + 2;
        const uint8_t *s = src[0];
        long ss = sstride[0];
        if (flip) { s += (long)(h - 1) * ss; ss = -ss; }
        uint8_t *y = dst[0], *u = dst[1], *v = dst[2];
        for (int r = 0; r < h; r += 2) {
            for (int x = 0; x < w; x += 2) {
                const uint8_t *q = s + 2 * x;
                y[x] = q[y1]; y[x + 1] = q[y2];
                u[x >> 1] = (uint8_t)((q[up] + q[up + ss] + 1) >> 1);
                v[x >> 1] = (uint8_t)((q[vp] + q[vp + ss] + 1) >> 1);
                y[dstride[0] + x] = q[ss + y1]; y[dstride[0] + x + 1] = q[ss + y2];
            }
            s += 2 * ss; y += 2 * dstride[0]; u += dstride[1]; v += dstride[2];
        }
        return 0;
    }
    case X264O_CSP_BGR: case X264O_CSP_BGRA: {
        const int step = id == X264O_CSP_BGR ? 3 : 4;
        uint32_t c[12];
        x264o_csp_rgb_coefs(colmatrix709, fullrange, c);
        const uint8_t *s = src[0];
        long ss = sstride[0];
        if (flip) { s += (long)(h - 1) * ss; ss = -ss; }
        uint8_t *y = dst[0], *u = dst[1], *v = dst[2];
        for (int r = 0; r < h; r += 2) {
            for (int x = 0; x < w; x += 2) {
                uint32_t cr = 0, cg = 0, cb = 0;
                for (int k = 0; k < 4; k++) {          /* (x,r) (x,r+1) (x+1,r) (x+1,r+1): the sums do not depend on the order */
                    const uint8_t *q = s + (long)(x + (k >> 1)) * step + (k & 1) * ss;
                    const uint32_t b = q[0], g = q[1], rr = q[2];
                    cr += rr; cg += g; cb += b;
                    y[(k & 1) * dstride[0] + x + (k >> 1)] = (uint8_t)((c[3] + c[0] * rr + c[1] * g + c[2] * b) >> BITS);
                }
                u[x >> 1] = (uint8_t)((c[7] + c[6] * cb - c[4] * cr - c[5] * cg) >> (BITS + 2));
                v[x >> 1] = (uint8_t)((c[11] + c[8] * cr - c[9] * cg - c[10] * cb) >> (BITS + 2));
            }
            s += 2 * ss; y += 2 * dstride[0]; u += dstride[1]; v += dstride[2];
        }
        return 0;
    }
    default: return -1;       /* NV12 / I422 / I444 / RGB targets are other encoder colourspaces (csp.c:489-512) */
    }
}

// common.hip.h — shared host/device declarations for the x264gpu library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/x264gpu.h"
#include "dsp.hip.h"
#include "dsp8.hip.h"

namespace x264gpu {

int set_err(int code, const char *what, hipError_t e);

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e__ = (expr);                                             \
        if (e__ != hipSuccess) return ::x264gpu::set_err(X264GPU_EHIP, #expr, e__); \
    } while (0)
#define ARG_TRY(cond)                                                        \
    do {                                                                     \
        if (!(cond)) return ::x264gpu::set_err(X264GPU_EINVAL, #cond, hipSuccess);  \
    } while (0)

// ---- host-side construction of the flat-CQM quant parameters (mirrors x264's cqm init; the three
//      parity classes are (even,even) / mixed / (odd,odd) coefficient positions) ----
struct QuantCfg { int deadzone_inter = 21, deadzone_intra = 11; };

inline Q4 make_q4(int qp, int list, const QuantCfg &c = QuantCfg())
{
    static const int qs[6][3] = { { 13107, 8066, 5243 }, { 11916, 7490, 4660 }, { 10082, 6554, 4194 },
                                  { 9362, 5825, 3647 },  { 8192, 5243, 3355 },  { 7282, 4559, 2893 } };
    static const int ds[6][3] = { { 10, 13, 16 }, { 11, 14, 18 }, { 13, 16, 20 },
                                  { 14, 18, 23 }, { 16, 20, 25 }, { 18, 23, 29 } };
    const int dz[4] = { 32 - c.deadzone_intra, 32 - c.deadzone_inter, 32 - 11, 32 - 21 };
    Q4 q;
    q.qp = qp;
    int sh = qp / 6 - 1;
    for (int k = 0; k < 3; k++) {
        int base = qs[qp % 6][k];
        int mf = sh <= 0 ? base << -sh : (base + (1 << (sh - 1))) >> sh;
        int b = ((dz[list] << 10) + (mf >> 1)) / mf, cap = (1 << 15) / mf;
        q.mf[k] = mf;
        q.bias[k] = b < cap ? b : cap;
        q.dq[k] = ds[qp % 6][k] * 16;
    }
    return q;
}

// 8x8 luma quantiser: six normAdjust8x8 position classes (list 0 intra, 1 inter)
inline Q8 make_q8(int qp, int list, const QuantCfg &c = QuantCfg())
{
    static const int qs[6][6] = { { 13107, 11428, 20972, 12222, 16777, 15481 }, { 11916, 10826, 19174, 11058, 14980, 14290 },
                                  { 10082, 8943, 15978, 9675, 12710, 11985 },   { 9362, 8228, 14913, 8931, 11984, 11259 },
                                  { 8192, 7346, 13159, 7740, 10486, 9777 },     { 7282, 6428, 11570, 6830, 9118, 8640 } };
    static const int ds[6][6] = { { 20, 18, 32, 19, 25, 24 }, { 22, 19, 35, 21, 28, 26 }, { 26, 23, 42, 24, 33, 31 },
                                  { 28, 25, 45, 26, 35, 33 }, { 32, 28, 51, 30, 40, 38 }, { 36, 32, 58, 34, 46, 43 } };
    const int dz = list == 0 ? 32 - c.deadzone_intra : 32 - c.deadzone_inter;
    Q8 q;
    q.qp = qp;
    const int sh = qp / 6;
    for (int k = 0; k < 6; k++) {
        int base = qs[qp % 6][k];
        int mf = sh <= 0 ? base : (base + (1 << (sh - 1))) >> sh;
        int b = ((dz << 10) + (mf >> 1)) / mf, cap = (1 << 15) / mf;
        q.mf[k] = mf; q.bias[k] = b < cap ? b : cap; q.dq[k] = ds[qp % 6][k] * 16;
    }
    return q;
}

inline int chroma_qp_of(int qp_luma, int offset)
{
    static const uint8_t tab[52] = { 0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16, 17,
                                     18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,
                                     34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };
    int q = qp_luma + offset;
    q = q < 0 ? 0 : q > 51 ? 51 : q;
    return tab[q];
}

inline int lambda_of(int qp)
{
    // max(1, round(2^(qp/6 - 2))) evaluated in exact integer arithmetic on 2^(1/6) steps
    static const double step[6] = { 1.0, 1.122462048309373, 1.2599210498948732, 1.4142135623730951,
                                    1.5874010519681994, 1.7817974362806785 };
    double v = step[qp % 6] * (double)(1 << (qp / 6)) / 4.0;
    int l = (int)(v + 0.5);
    return l < 1 ? 1 : l;
}

}  // namespace x264gpu

// mb_slice_intra.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for I slices: no motion search, so one instantiation serves
// every --me method and sub-pel margin.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_intra(const EncK &k, int streams, hipStream_t st);        // mb_slice_ref_intra.hip
void launch_mb_slice_intra(const EncK &k, int streams, hipStream_t st)
{
    if (((k.rd >> 1) & 63) && k.cabac) { launch_mb_slice_ref_intra(k, streams, st); return; }      // RD refinement (subme 8)
    if (k.rd && k.cabac && (k.trellis & 64)) hipLaunchKernelGGL((k_mb_slice<2, 1, false, 4>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
    else if (k.rd && k.cabac && k.trellis) hipLaunchKernelGGL((k_mb_slice<2, 1, false, 3>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
    else if (k.rd && k.cabac) hipLaunchKernelGGL((k_mb_slice<2, 1, false, 2>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
    else if (k.rd) hipLaunchKernelGGL((k_mb_slice<2, 1, false, 1>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
    else hipLaunchKernelGGL((k_mb_slice<2, 1, false>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
}
}  // namespace x264gpu

// mb_slice_intra.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for I slices: no motion search, so one instantiation serves
// every --me method and sub-pel margin.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_intra(const EncK &k, int streams, hipStream_t st);        // mb_slice_ref_intra.hip
void launch_mb_slice_intra(const EncK &k, int streams, hipStream_t st)
{
    if (((k.rd >> 1) & 63) && k.cabac) { launch_mb_slice_ref_intra(k, streams, st); return; }      // RD refinement (subme 8)
    if (k.rd && k.cabac && (k.trellis & 64)) mb_launch(k_mb_slice<2, 1, false, 4>, k, streams, st);
    else if (k.rd && k.cabac && k.trellis) mb_launch(k_mb_slice<2, 1, false, 3>, k, streams, st);
    else if (k.rd && k.cabac) mb_launch(k_mb_slice<2, 1, false, 2>, k, streams, st);
    else if (k.rd) mb_launch(k_mb_slice<2, 1, false, 1>, k, streams, st);
    else mb_launch(k_mb_slice<2, 1, false>, k, streams, st);
}
}  // namespace x264gpu

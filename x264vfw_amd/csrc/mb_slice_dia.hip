// mb_slice_dia.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for --me dia: one translation unit per search method so that
// the four instantiation pairs build in parallel.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_dia(const EncK &k, int streams, bool big_margin, hipStream_t st)
{
    if (k.rd && k.cabac && (k.trellis & 64)) mb_launch(k_mb_slice<2, 0, true, 4>, k, streams, st);      // ... --trellis 2
    else if (k.rd && k.cabac && k.trellis) mb_launch(k_mb_slice<2, 0, true, 3>, k, streams, st);      // ... and trellis
    else if (k.rd && k.cabac) mb_launch(k_mb_slice<2, 0, true, 2>, k, streams, st);      // ... with CABAC sizes
    else if (k.rd) mb_launch(k_mb_slice<2, 0, true, 1>, k, streams, st);      // RD: subme 6 / 7, the +-2 sub-pel neighbourhood
    else if (big_margin) mb_launch(k_mb_slice<5, 0, true>, k, streams, st);
    else mb_launch(k_mb_slice<2, 0, true>, k, streams, st);
}
}  // namespace x264gpu

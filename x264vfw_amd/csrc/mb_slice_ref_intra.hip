// mb_slice_ref_intra.hip — the macroblock-loop kernel (k_mb.hip.h) for I slices with RD refinement of the intra modes (x264 --subme 8: intra_rd_refine,
// k_mb_refine.inc), trellis 0 / 1 (RD 5) and trellis 2 (RD 6).
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_intra(const EncK &k, int streams, hipStream_t st)
{
    if (k.trellis & 64) mb_launch(k_mb_slice<2, 1, false, 6>, k, streams, st);
    else mb_launch(k_mb_slice<2, 1, false, 5>, k, streams, st);
}
}  // namespace x264gpu

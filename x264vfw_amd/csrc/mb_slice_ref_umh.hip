// mb_slice_ref_umh.hip — the macroblock-loop kernel (k_mb.hip.h) with RD refinement (x264 --subme 8: k_mb_refine.inc) for P slices under --me umh: the
// +-5 sample sub-pel neighbourhood (4 half-pel + 10 quarter-pel iterations), trellis 0 / 1 (RD 5) and trellis 2 (RD 6); a translation unit of its own.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_umh(const EncK &k, int streams, hipStream_t st)
{
    if (k.trellis & 64) mb_launch(k_mb_slice<5, 2, true, 6>, k, streams, st);
    else mb_launch(k_mb_slice<5, 2, true, 5>, k, streams, st);
}
}  // namespace x264gpu

// mb_slice_ref_b_hex.hip — the macroblock-loop kernel (k_mb.hip.h) for B slices at x264 --subme 9 (i_mbrd 2): RD refinement of the decision's winner
// (k_mb_refine.inc behind k_mb_b.inc), --me hex, trellis 0 / 1 (RD 5) and trellis 2 (RD 6), the +-5 sample sub-pel
// neighbourhood (B slices search with 4 half-pel + 10 quarter-pel iterations from this level on); a translation unit of its own.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_b_hex(const EncK &k, int streams, hipStream_t st)
{
    if (k.trellis & 64) mb_launch(k_mb_slice<5, 1, true, 6, true>, k, streams, st);
    else mb_launch(k_mb_slice<5, 1, true, 5, true>, k, streams, st);
}
}  // namespace x264gpu

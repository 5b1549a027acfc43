// mb_slice_b_umh.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for B slices under --me umh (RD sessions with CABAC; with and without
// the trellis quantiser): a translation unit of its own so that the instantiations build in parallel.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_ref_b_umh(const EncK &k, int streams, hipStream_t st);        // mb_slice_ref_b_umh.hip
void launch_mb_slice_b1_umh(const EncK &k, int streams, hipStream_t st);        // mb_slice_b1_umh.hip
void launch_mb_slice_b0_umh(const EncK &k, int streams, hipStream_t st);        // mb_slice_b0_umh.hip
void launch_mb_slice_b_umh(const EncK &k, int streams, hipStream_t st)
{
    if (!k.rd || k.subme < 7) { launch_mb_slice_b0_umh(k, streams, st); return; }
    if (!k.cabac) { launch_mb_slice_b1_umh(k, streams, st); return; }      // RD with CAVLC bit counts
    if (k.cabac && (k.subme >= 9 || (k.rd & 64))) { launch_mb_slice_ref_b_umh(k, streams, st); return; }      // --subme 9: the +-5 sample sub-pel neighbourhood (4 + 10 iterations) and RD refinement of the sites cfg.rd names
    if (k.trellis & 64) mb_launch(k_mb_slice<2, 2, true, 4, true>, k, streams, st);      // --trellis 2
    else if (k.trellis) mb_launch(k_mb_slice<2, 2, true, 3, true>, k, streams, st);
    else mb_launch(k_mb_slice<2, 2, true, 2, true>, k, streams, st);
}
}  // namespace x264gpu

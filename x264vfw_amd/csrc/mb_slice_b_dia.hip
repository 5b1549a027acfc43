// mb_slice_b_dia.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for B slices under --me dia (RD sessions with CABAC; with and without
// the trellis quantiser): a translation unit of its own so that the instantiations build in parallel.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_b1_dia(const EncK &k, int streams, hipStream_t st);        // mb_slice_b1_dia.hip
void launch_mb_slice_b0_dia(const EncK &k, int streams, hipStream_t st);        // mb_slice_b0_dia.hip
void launch_mb_slice_b_dia(const EncK &k, int streams, hipStream_t st)
{
    if (!k.rd || k.subme < 7) { launch_mb_slice_b0_dia(k, streams, st); return; }
    if (!k.cabac) { launch_mb_slice_b1_dia(k, streams, st); return; }      // RD with CAVLC bit counts      // B slices below --subme 7: analysis without RD
    if (k.trellis & 64) mb_launch(k_mb_slice<2, 0, true, 4, true>, k, streams, st);      // --trellis 2
    else if (k.trellis) mb_launch(k_mb_slice<2, 0, true, 3, true>, k, streams, st);
    else mb_launch(k_mb_slice<2, 0, true, 2, true>, k, streams, st);
}
}  // namespace x264gpu

// mb_slice_b0_esa.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for B slices analysed without RD (x264 below --subme 7 in B slices:
// probe_bskip, SATD decisions, me_refine_qpel of the winner; k_mb_b.inc's RD == 0 branches), --me esa.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_b0_esa(const EncK &k, int streams, hipStream_t st)
{
    // (with the trellis quantiser the final encode reads the slice's CABAC state: the instantiation that carries it)
    if (k.rd && k.cabac && k.trellis) mb_launch(k_mb_slice<2, 3, true, 7, true>, k, streams, st);
    else mb_launch(k_mb_slice<2, 3, true, 0, true>, k, streams, st);
}
}  // namespace x264gpu

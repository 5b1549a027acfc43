// mb_slice_b0_esa.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for B slices analysed without RD (x264 below --subme 7 in B slices:
// probe_bskip, SATD decisions, me_refine_qpel of the winner; k_mb_b.inc's RD == 0 branches), --me esa.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_b0_esa(const EncK &k, int streams, hipStream_t st)
{
    // (with the trellis quantiser the final encode reads the slice's CABAC state: the instantiation that carries it)
    if (k.rd && k.cabac && k.trellis) hipLaunchKernelGGL((k_mb_slice<2, 3, true, 7, true>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
    else hipLaunchKernelGGL((k_mb_slice<2, 3, true, 0, true>), dim3(streams, k.slices > 1 ? k.slices : 1), dim3(64), 0, st, k);
}
}  // namespace x264gpu

// mb_slice_b1_hex.hip — the macroblock-loop kernel (k_mb.hip.h) instantiated for B slices of RD sessions WITHOUT CABAC (x264 --no-cabac at --subme 7 and
// up: the B decisions priced with CAVLC bit counts, cavlc_mb_header_b as a count), --me hex; a translation unit of its own.
#include "k_mb.hip.h"

namespace x264gpu {
void launch_mb_slice_b1_hex(const EncK &k, int streams, hipStream_t st)
{
    mb_launch(k_mb_slice<2, 1, true, 1, true>, k, streams, st);
}
}  // namespace x264gpu

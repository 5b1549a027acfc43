"""Where the macroblock loop's cycles go: phase counters of k_mb_slice (a -DMB_PROF build of the device library).

    make -C x264vfw_amd/csrc clean && make -C x264vfw_amd/csrc -j8 EXTRA=-DMB_PROF && python tools/mb_prof.py [streams] [frames]

Prints, per picture, the average cycles per macroblock of every phase (s_memtime deltas summed by lane 0 of each stream's wavefront).
The timers perturb scheduling a little; use the table for proportions, bench.py for absolute time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from x264vfw_amd import lib  # noqa: E402
from x264vfw_amd.lib import Config, MB_LEVELS  # noqa: E402

PH = ["setup", "me_pred", "me_win", "me_fpel", "me_substage", "me_subpel", "me_glue", "pskip", "intra_chroma", "intra", "enc_inter", "enc_intra", "store",
      "n_search", "n_refine", "n_stage"]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
    H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
    dev = torch.device("cuda:0")
    args = type("A", (), dict(refs=3, preset="medium", aq=False, rd=os.environ.get("MB_PROF_RD", "cabac"), no_trellis=bool(os.environ.get("MB_PROF_NO_TRELLIS")),
                              bframes=int(os.environ.get("MB_PROF_BFRAMES", "3"))))()
    tools = bench.toolset(args)
    D = min(S, 64)
    base = bench.synth_batch(torch, D, F, W, H, 0x264, dev)
    data = base if D == S else base.repeat(1, (S + D - 1) // D, 1)[:, :S].contiguous()
    cfg = Config(**dict(dict(width=W, height=H, streams=S, qp_i=20, qp_p=23, me_range=16, deblock_alpha=0, deblock_beta=0, chroma_qp_offset=0, deadzone_inter=21,
                             deadzone_intra=11, dct_decimate=1), **tools))
    h = C.c_void_p()
    lib.check(lib.x264gpu_encoder_create(C.byref(h), C.byref(cfg)), "create")
    n = lib.x264gpu_encoder_mb_count(h)
    mbs = torch.empty((S, n, 64), dtype=torch.uint8, device=dev)
    lvs = torch.empty((S, n, MB_LEVELS), dtype=torch.int16, device=dev)
    f = lib._lib.x264gpu_encoder_mb_prof
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros((S, 32), dtype=np.uint64)
    print("%-4s %10s " % ("pic", "cyc/MB") + " ".join("%9s" % p[:9] for p in PH[:16]))
    from x264vfw_amd import gop, host_api as HL
    from x264vfw_amd.lib import Pic
    bfr = int(os.environ.get("MB_PROF_BFRAMES", "3")) if tools.get("dpb") else 0
    order = gop.schedule(bench.display_types(F, bfr, F), 1)
    dpb = gop.HostDpb(HL, tools["refs"], bfr, 1, weightp=2)
    qargs = type("Q", (), dict(qp=23))()
    for i, (disp, pt) in enumerate(order):
        pic, _ = dpb.plan(pt, disp, gop.follow_of(order, i))
        pic.qp = bench.qp_of(qargs, pt)
        dpb.commit()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.check(lib.x264gpu_encode_pictures(h, data[disp].data_ptr(), (Pic * S)(*([pic] * S)), mbs.data_ptr(), lvs.data_ptr(), torch.cuda.current_stream().cuda_stream), "encode")
        e1.record()
        torch.cuda.synchronize()
        lib.check(f(h, out.ctypes.data), "mb_prof (is this an MB_PROF build?)")
        a = out.astype(np.float64).mean(axis=0) / n
        tot = a[:13].sum() + (a[15] if a[15] > 100 else 0)          # -DMB_PROF_RD builds: slot 15 = cycles of the CABAC pricing (RD sessions)
        print("%-4s %10.0f " % ("IIPRb"[pt] + str(disp), tot) + " ".join("%9.0f" % v for v in a[:13]) + " %9.2f %9.2f %9.2f" % (a[13], a[14], a[15]) + "   %.1f ms" % e0.elapsed_time(e1))
        if a[16] > 0:         # -DMB_PROF_RD: inside the CABAC pricing (cabac_rd.hip.h), per macroblock
            print("     cab_mb calls %.2f  header %.0f  cbf+sigmaps %.0f  levels: prep %.0f walk %.0f rest %.0f cycles;  walk steps %.1f  non-zero coefficients %.1f" % (a[16], a[17], a[18], a[22], a[23], a[19], a[20], a[21]))
        if a[27] > 0:
            print("     around the pricing: candidate's distortion terms + header inputs %.0f  cost bookkeeping %.0f cycles" % (a[27], a[28]))
        if a[24] > 0:
            print("     inter encode: prediction (b_predict) %.0f  luma transform / quantiser / reconstruction %.0f  chroma %.0f cycles" % (a[24], a[25], a[26]))
        mx = out[:, :13].sum(axis=1).astype(np.float64)
        print("     slowest/mean stream cycles: %.3f   share: " % (mx.max() / mx.mean()) + " ".join("%8.1f%%" % (100 * v / tot) for v in a[:13]))


if __name__ == "__main__":
    main()

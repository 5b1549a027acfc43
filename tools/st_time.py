"""Times x264gpu_slicetype_frame_cost (the lookahead's frame costs, csrc/slicetype.hip k_st_cost) for a batch of streams:
    python tools/st_time.py [streams] [width] [height]
P costs at distances 1 / 2 and B costs between two references, each on fresh pictures (a cost is cached once computed)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from x264vfw_amd import lib
from x264vfw_amd.synth import synth_frames

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
nfr, nslots = 6, 8
fr = synth_frames(W, H, nfr * 2, seed=11)
data = [torch.from_numpy(np.stack([fr[(i + (s % 2) * nfr) % len(fr)] for s in range(S)])).cuda() for i in range(nfr)]
st = C.c_void_p()
lib.check(lib.x264gpu_slicetype_create(C.byref(st), W, H, S, nslots, 3, 1, 7, 16, 1, 512, 1), "create")
score = np.zeros(S, np.int32)
res = {}
for rep in range(2):
    for i in range(nfr):
        lib.check(lib.x264gpu_slicetype_put_frame(st, i, data[i].data_ptr(), None), "put")
    torch.cuda.synchronize()
    def cost(tag, p0, p1, b):
        t = time.perf_counter()
        lib.check(lib.x264gpu_slicetype_frame_cost(st, p0, p1, b, b - p0, p1 - b, score.ctypes.data, None), "cost")
        torch.cuda.synchronize()
        res.setdefault(tag, []).append((time.perf_counter() - t) * 1e3)
    cost("I", 0, 0, 0); cost("P d1", 0, 1, 1); cost("P d2", 0, 2, 2); cost("B (0,2,1)", 0, 2, 1); cost("P d1 b", 2, 3, 3); cost("B (2,5,3) one list searched", 2, 5, 3); cost("B (2,5,4)", 2, 5, 4)
print(f"{S} streams {W}x{H}: " + "; ".join(f"{k} {min(v):.1f} ms" for k, v in res.items()), "| checksum", int(score.astype(np.int64).sum()))
lib.x264gpu_slicetype_destroy(st)

import sys, time
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from x264vfw_amd import lib
rng = np.random.default_rng(1)
for cat, nc in ((2, 16), (5, 64), (4, 16), (3, 4)):
    nblk = 8000
    amp = 400.0 / (1.0 + 0.35 * np.arange(nc))
    coefs = (rng.laplace(0, 1, (nblk, nc)) * amp).astype(np.int16)
    if cat == 4: coefs[:, 0] = 0
    states = ((rng.integers(0, 63, 460) << 1) | rng.integers(0, 2, 460)).astype(np.uint8)
    d_c, d_s = torch.from_numpy(coefs).cuda(), torch.from_numpy(states).cuda()
    d_l, d_z = torch.zeros_like(d_c), torch.zeros(nblk, dtype=torch.uint8, device='cuda')
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        lib.check(lib.x264gpu_trellis_blocks(d_c.data_ptr(), nblk, cat, 23, 0, d_s.data_ptr(), d_l.data_ptr(), d_z.data_ptr(), None), "t")
        torch.cuda.synchronize(); dt = time.time() - t0
    nnz = np.count_nonzero(d_l.cpu().numpy()) / nblk
    print(f"cat {cat}: {dt*1e6/(nblk/8):.1f} us per pass of 8 blocks = {dt*1e6/(nblk/8)*2400/ (nc - (1 if cat==4 else 0)):.0f} cycles per step (upper bound), {nnz:.1f} nonzero levels per block")

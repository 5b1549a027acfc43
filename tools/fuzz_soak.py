"""Longer randomised parity soak than tests/test_gpu_fuzz.py runs per round: python tools/fuzz_soak.py SEED0 SEED1 [cases per seed] [rd | rdcabac | trellis | slices].
With "rd" / "rdcabac" every case runs with RD mode decision on (CAVLC / CABAC session, subme 6 / 7, random psy-RD strength).
"slices": every case with x264 --slices N (slices_plain, 2 .. one per macroblock row) on content with scene cuts every few pictures, so that
P pictures hold intra macroblocks and the repeated slice passes have work; a third of the cases with RD + CABAC + trellis.
Every random case (tests/test_gpu_fuzz.py random_case) is encoded by the HIP pipeline and the oracle; mismatches are listed, not fatal."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from gpu_enc import GpuEncoder  # noqa: E402
from synth import synth_frames  # noqa: E402
from test_gpu_fuzz import random_case  # noqa: E402


def main():
    s0, s1 = int(sys.argv[1]), int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    rd = len(sys.argv) > 4 and sys.argv[4] in ("rd", "rdcabac", "trellis")
    rd_cabac = len(sys.argv) > 4 and sys.argv[4] in ("rdcabac", "trellis")
    trellis = len(sys.argv) > 4 and sys.argv[4] == "trellis"
    plain = len(sys.argv) > 4 and sys.argv[4] == "slices"
    bad = total = 0
    t0 = time.time()
    for seed in range(s0, s1):
        rnd = random.Random(seed)
        for it in range(per):
            w, h, kw, nfr, fseed, second_idr = random_case(rnd)
            if rd:
                psy = rnd.randint(0, 1)
                if trellis:
                    kw.update(trellis=rnd.choice([63, 63, 127, rnd.randint(1, 62), 64 + rnd.randint(1, 63)]))
                kw.update(cabac=int(rd_cabac), rd=1, subme=rnd.choice([6, 7]), psy=psy, psy_rd_q8=rnd.choice([26, 102, 256, 512]) if psy else 0)
            scene_len = 97
            if plain:
                h = max(h, 32)
                kw.update(slices=rnd.randint(2, (h + 15) // 16), slices_plain=1)
                scene_len = rnd.choice([2, 3, 4, 97])
                nfr = max(nfr, 4)
                if rnd.random() < 0.33:
                    kw.update(cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, trellis=63)
            frames = synth_frames(w, h, nfr, seed=fseed, scene_len=scene_len)
            cfg = O.default_config(w, h, **kw)
            og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
            total += 1
            for i, f in enumerate(frames):
                st = 2 if i == 0 or (i == 3 and second_idr) else 0
                o_mb, o_lv = og.encode(f, st)
                g_mb, g_lv = gg.encode([f], st)
                ok = np.array_equal(g_mb[0].view(np.uint8), o_mb.view(np.uint8)) and np.array_equal(g_lv[0], o_lv) and np.array_equal(gg.recon(0), og.recon())
                if not ok:
                    bad += 1
                    d = np.nonzero((g_mb[0].view(np.uint8).reshape(-1, 64) != o_mb.view(np.uint8).reshape(-1, 64)).any(axis=1))[0]
                    print(f"MISMATCH seed {seed} case {it}: {w}x{h} {kw} frame {i}: {len(d)} records differ, first MB {d[:1]}", flush=True)
                    break
            og.close(); gg.close()
    print(f"{total} cases, {bad} mismatching, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

"""Randomised parity soak of the B-picture path (and --weightp 2): python tools/fuzz_soak_b.py SEED0 SEED1 [cases per seed].
Every case draws a picture size, a string of picture types (runs of 0..3 B pictures, with and without b-pyramid), a toolset around preset medium
(references, search method and range, partitions with p8x8 / b8x8 apart, 8x8 transform, mixed references, weightb, trellis 0 / 1 / 2, psy strength,
deblocking, slices as x264's slice threads or --slices N, the blind duplicate of --weightp 2) and quantisers; tests/test_gpu_bframes.py run()
codes it on the device and on the CPU checker (records, levels, reconstruction, context variables) and decodes the device's stream.  Mismatches
are listed, not fatal."""
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_b_case(rnd):
    w = rnd.choice([64, 96, 128, 176, 208, 200, 72])
    h = rnd.choice([48, 80, 96, 144, 112, 104, 136])
    if os.environ.get("FUZZ_BIG"):          # a slower soak on larger pictures
        w, h = rnd.choice([(352, 288), (416, 240), (320, 192), (480, 272)])
    bframes = rnd.randint(1, 3)
    types = "I"
    while len(types) < rnd.randint(4, 9):
        types += "B" * rnd.randint(0, bframes) + "P"
    refs = rnd.randint(1, 4)
    pyramid = rnd.randint(0, 1) if bframes > 1 else 0
    kw = dict(refs=refs, dpb=max(refs, 4 if pyramid else 2, 2), weightb=rnd.randint(0, 1), mixed_refs=rnd.randint(0, 1), dct8x8=rnd.randint(0, 1),
              me_method=rnd.choice([0, 1, 1, 2, 3]), me_range=rnd.choice([4, 8, 16]), trellis=rnd.choice([0, 63, 63, 127]), deblock=rnd.randint(0, 1) or 1,
              fast_pskip=rnd.randint(0, 1) or 1, dct_decimate=rnd.randint(0, 1) or 1)
    kw["partitions"] = rnd.choice([7, 7, 6, 0x707, 0xf06, 0xf07, 3, 5])
    if not kw["dct8x8"]:
        kw["partitions"] &= ~0x404
    psy = rnd.randint(0, 1)
    kw.update(psy=psy, psy_rd_q8=rnd.choice([26, 102, 256, 512]) if psy else 0, chroma_qp_offset=rnd.choice([0, -2, -1, 3]) if psy else rnd.choice([0, 2]))
    if rnd.random() < 0.3:
        mbh = (h + 15) // 16
        if rnd.random() < 0.5 and mbh >= 8:
            kw.update(slices=rnd.randint(2, mbh // 4))
        else:
            kw.update(slices=rnd.randint(2, mbh), slices_plain=1)
    if rnd.random() < 0.3:           # variance AQ: per-macroblock quantisers, the within-1 rule of x264_macroblock_analyse, mb_qp_delta in the RD costs
        kw.update(aq_mode=1, aq_strength=rnd.choice([0.51985, 1.0397, 1.55955]))
        if rnd.random() < 0.6:
            kw["_qp_frac"] = [rnd.randint(-128, 127) for _ in range(4)]
    weightp = rnd.choice([0, 0, 2]) if refs >= 2 else 0
    if rnd.random() < 0.25:          # explicit luma weights on the P pictures (what x264_weights_analyse hands a fade): --weightp 1 or 2
        weightp = rnd.choice([1, 2])
        wts = {}
        for i, t in enumerate(types):
            if t == "P" and rnd.random() < 0.8:
                denom = rnd.randint(0, 7)
                scale = max(1, min(127, (1 << denom) + rnd.randint(-(1 << denom) // 4 - 1, (1 << denom) // 4 + 1))) if rnd.random() < 0.8 else rnd.randint(1, 127)
                wts[i] = (scale, denom, rnd.randint(-12, 12) if rnd.random() < 0.8 else rnd.randint(-128, 127))
        kw["_weights"] = wts
    if rnd.random() < 0.15:          # sessions without RD (subme <= 5) run P pictures only on the DPB model: duplicates and weights there
        types = types.replace("B", "P")
        kw.update(rd=0, subme=rnd.randint(1, 5), trellis=0, psy=0, psy_rd_q8=0, chroma_qp_offset=rnd.choice([0, 2]), cabac=rnd.randint(0, 1))
        if "_weights" in kw:
            kw["_weights"] = {i: v for i, v in kw["_weights"].items() if types[i] == "P"}
    if os.environ.get("FUZZ_R04") and kw.get("rd", 1) and "B" in types:
        # round 4's additions (drawn after the draws above: the committed seeds of tests/test_gpu_fuzz.py stay what they were): the levels of --subme in
        # sessions with B pictures (B analysis without RD below 7, RD refinement 8 / 9 with its sites, deblock-aware RD), the three --direct modes
        lvl = rnd.choice([9, 9, 9, 8, 8, 7, 6, 5, 4, 3, 2, 1])
        if lvl >= 8:
            if kw["me_method"] not in (1, 2):
                kw["me_method"] = rnd.choice([1, 2])
            kw.update(subme=lvl, rd=rnd.choice([63, 63, 63 | 64, 63 | 64, 3, 61, 1 | 64, 1, 17 | 64, 9, 37]))
        elif lvl == 7 and rnd.random() < 0.3:          # --no-cabac at subme 7: the B decisions on CAVLC bit counts
            kw.update(subme=7, rd=1, cabac=0, trellis=0)
        elif lvl == 7:
            kw.update(subme=7, rd=rnd.choice([1, 1 | 64]) if kw["me_method"] in (1, 2) else 1)          # (deblock-aware RD lives in the refinement instantiations: hex / umh)
        elif lvl == 6:
            kw.update(subme=6, rd=1)
        else:
            kw.update(subme=lvl, rd=0, trellis=0, psy=0, psy_rd_q8=0, chroma_qp_offset=rnd.choice([0, 2]), cabac=rnd.randint(0, 1))
        kw["direct"] = rnd.choice(["spatial", "spatial", "temporal", "auto", "auto"])
        if os.environ.get("FUZZ_BIG"):
            w, h = rnd.choice([(352, 288), (416, 240), (640, 360), (480, 272)])
            if "slices" in kw:          # (drawn for the first size)
                mbh = (h + 15) // 16
                kw["slices"] = max(2, min(kw["slices"], mbh if kw.get("slices_plain") else mbh // 4))
    return w, h, types, rnd.randint(1, 10 ** 6), bframes, pyramid, weightp, kw


def panned_frames(w, h, n, seed, dx, dy):
    """a fast global pan (dx, dy pixels a picture) over one synthetic picture: long vectors, also across the picture borders"""
    import numpy as np
    from synth import synth_frames
    base = synth_frames(w, h, 1, seed=seed)[0]
    y = base[:w * h].reshape(h, w); u = base[w * h:w * h * 5 // 4].reshape(h // 2, w // 2); v = base[w * h * 5 // 4:].reshape(h // 2, w // 2)
    out = []
    for i in range(n):
        sx, sy = (i * dx) // 2 * 2, (i * dy) // 2 * 2
        out.append(np.concatenate([np.roll(y, (sy, sx), (0, 1)).ravel(), np.roll(u, (sy // 2, sx // 2), (0, 1)).ravel(), np.roll(v, (sy // 2, sx // 2), (0, 1)).ravel()]))
    return out


def main():
    import pytest  # noqa: F401  (tests/conftest fixtures are not used: run() takes the gpu argument for show)
    from test_gpu_bframes import run
    s0, s1 = int(sys.argv[1]), int(sys.argv[2])
    per = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    bad = total = 0
    t0 = time.time()
    for seed in range(s0, s1):
        rnd = random.Random(seed)
        for it in range(per):
            w, h, types, fseed, bframes, pyramid, weightp, kw = random_b_case(rnd)
            total += 1
            try:
                frames = panned_frames(w, h, len(types), fseed, rnd.randint(-30, 30), rnd.randint(-30, 30)) if rnd.random() < 0.25 else None
                run(None, w, h, types, fseed, bframes=bframes, pyramid=pyramid, weightp=weightp, weights=kw.pop("_weights", None), frames=frames, qp_frac=kw.pop("_qp_frac", None), **kw)
            except AssertionError as e:
                bad += 1
                print(f"MISMATCH seed {seed} case {it}: {w}x{h} {types} bframes {bframes} pyramid {pyramid} weightp {weightp} {kw}: {str(e)[:300]}", flush=True)
            except Exception:
                bad += 1
                print(f"ERROR seed {seed} case {it}: {w}x{h} {types} bframes {bframes} pyramid {pyramid} weightp {weightp} {kw}", flush=True)
                traceback.print_exc()
    print(f"{total} cases, {bad} bad, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

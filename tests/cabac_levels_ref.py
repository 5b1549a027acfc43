"""Serial restatement of the residual coding (significance maps of 4x4-type blocks + coeff_abs_level_minus1) of a macroblock's blocks as x264's size-only CABAC coder prices it
([x264-upstream] encoder/cabac.c residual_block_cabac / encoder/rdo.c: coeff_abs_level1_ctx, coeff_abs_levelgt1_ctx, coeff_abs_level_transition), block
after block and bin after bin — the checker of the device's all-blocks-at-once level walk (csrc/cabac_rd.hip.h cab_levels_all, primitive
x264gpu_cabac_level_walk).  Test infrastructure: pure Python, the entropy table is the generated csrc/cabac_entropy.inc (tools/gen_cabac_entropy.py)."""
import os
import random
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ENT = [int(x) for x in re.findall(r"\d+", "".join(l for l in open(os.path.join(_HERE, "..", "x264vfw_amd", "csrc", "cabac_entropy.inc")) if not l.startswith("//")))]
assert len(_ENT) == 128
TRANS_LPS = [0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
             24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63]
LV_LUMA_DC, LV_CHROMA_DC, LV_CHROMA_AC, MB_LEVELS = 256, 272, 280, 416          # include/x264gpu.h


def step(st, b):
    """one bin on context variable st = (pStateIdx << 1) | valMPS: the variable after it, its cost in 1/256 bit"""
    sg, mps = st >> 1, st & 1
    lps = (mps ^ b) != 0
    cost = _ENT[2 * sg + 1] if lps else _ENT[2 * sg]
    ns = TRANS_LPS[sg] if lps else min(sg + 1, 62)
    nm = mps ^ 1 if (lps and sg == 0) else mps
    return (ns << 1) | nm, cost


def block_levels(coefs, ctx, bits):
    """the levels of one block (scan order), last coefficient first, on the category's ten coeff_abs_level_minus1 contexts ctx (updated)"""
    node = 0
    c1t, cgt = [1, 2, 3, 4, 0, 0, 0, 0], [5, 5, 5, 5, 6, 7, 8, 9]
    tr0, tr1 = [1, 2, 3, 3, 4, 5, 6, 7], [4, 4, 4, 4, 5, 6, 7, 7]
    for i in range(len(coefs) - 1, -1, -1):
        a = abs(int(coefs[i]))
        if not a:
            continue
        c = c1t[node]
        if a > 1:
            ctx[c], co = step(ctx[c], 1); bits += co
            c = cgt[node]
            for _ in range(min(a, 15) - 2):
                ctx[c], co = step(ctx[c], 1); bits += co
            if a < 15:
                ctx[c], co = step(ctx[c], 0); bits += co
            else:
                bits += 256 * (2 * ((a - 15 + 1).bit_length() - 1) + 1)          # Exp-Golomb order 0 suffix, bypass bins
            node = tr1[node]
        else:
            ctx[c], co = step(ctx[c], 0); bits += co
            node = tr0[node]
        bits += 256                                                              # sign
    return bits


def block_sigmap(coefs, sig, last, bits):
    """significant_coeff_flag / last_significant_coeff_flag of one 4x4-type block (scan order; position i has contexts sig[i] / last[i]; the last position
    of a block has neither) — 7.3.5.3.3 residual_block_cabac up to the levels"""
    nz = [i for i, c in enumerate(coefs) if c]
    lastp = nz[-1]
    for i in range(lastp + 1):
        if i >= len(coefs) - 1:
            break
        sig[i], co = step(sig[i], 1 if coefs[i] else 0); bits += co
        if coefs[i]:
            last[i], co = step(last[i], 1 if i == lastp else 0); bits += co
    return bits


def _block(coefs, r0, byte, sigbase, lastbase, absbase, nabs, bits):
    """one 4x4-type block on the role-indexed register bytes r0[lane][byte]: significance map, then levels"""
    n1 = len(coefs) - 1
    sig = [r0[sigbase + i][byte] for i in range(n1)]; last = [r0[lastbase + i][byte] for i in range(n1)]
    bits = block_sigmap(coefs, sig, last, bits)
    for i in range(n1): r0[sigbase + i][byte] = sig[i]; r0[lastbase + i][byte] = last[i]
    ctx = [r0[absbase + i][byte] for i in range(nabs)] + [0] * (10 - nabs)
    bits = block_levels(coefs, ctx, bits)
    for i in range(nabs): r0[absbase + i][byte] = ctx[i]
    return bits


def random_case(rnd):
    """levels of one macroblock (x264gpu_mb layout), what to code of it, the role-indexed context registers r (4 bytes a lane) and r8 -> the
    expected registers and bits"""
    kind = rnd.choice([2, 5, 1, -1])
    scale = rnd.choice([1, 1, 2, 4, 20, 60])
    dens = rnd.random()
    lev = lambda p0: 0 if rnd.random() > p0 else rnd.choice([-1, 1]) * max(1, int(abs(rnd.gauss(0, scale))))
    lv = [0] * MB_LEVELS
    for i in range(256): lv[i] = lev(dens)
    for i in range(128): lv[LV_CHROMA_AC + i] = lev(dens * 0.6)
    for i in range(8): lv[LV_CHROMA_DC + i] = lev(0.7)
    for i in range(16): lv[LV_LUMA_DC + i] = lev(0.8)
    for b in range(8): lv[LV_CHROMA_AC + b * 16] = rnd.randint(-3, 3)            # whatever lies in the DC slot of AC blocks is not coded
    if kind == 1:
        for b in range(16): lv[b * 16] = rnd.randint(-3, 3)
    r = [[rnd.randrange(126) for _ in range(4)] for _ in range(64)]
    r8 = [rnd.randrange(126) for _ in range(64)]
    r0, r80 = [x[:] for x in r], r8[:]
    what = dict(cat0=kind, nz0=0, nzac=0, nzdc=0, ldc=0)
    bits = 0
    if kind == 1 and any(lv[LV_LUMA_DC:LV_LUMA_DC + 16]) and rnd.random() < 0.9:
        bits = _block(lv[LV_LUMA_DC:LV_LUMA_DC + 16], r0, 3, 0, 16, 32, 10, bits)
        what["ldc"] = 1
    if kind in (1, 2):
        byte = 1 if kind == 1 else 0
        for b in range(16):
            co = lv[b * 16 + (1 if kind == 1 else 0):b * 16 + 16]
            if any(co) and rnd.random() < 0.85:
                bits = _block(co, r0, byte, 0, 16, 32, 10, bits); what["nz0"] |= 1 << b
    if kind == 5:
        # (an 8x8 block's significance map is coded by cab_block8 in the macroblock layer, not by the walk: levels only here)
        ctx = [r80[32 + i] for i in range(10)]
        for b in range(4):
            co = [lv[(b * 4 + (p & 3)) * 16 + (p >> 2)] for p in range(64)]
            if any(co) and rnd.random() < 0.85:
                bits = block_levels(co, ctx, bits); what["nz0"] |= 1 << b
        for i in range(10): r80[32 + i] = ctx[i]
    for pl in range(2):
        co = lv[LV_CHROMA_DC + pl * 4:LV_CHROMA_DC + pl * 4 + 4]
        if any(co) and rnd.random() < 0.9:
            bits = _block(co, r0, 3, 48, 52, 55, 9, bits); what["nzdc"] |= 1 << pl
    for b in range(8):
        co = lv[LV_CHROMA_AC + b * 16 + 1:LV_CHROMA_AC + b * 16 + 16]
        if any(co) and rnd.random() < 0.85:
            bits = _block(co, r0, 2, 0, 16, 32, 10, bits); what["nzac"] |= 1 << b
    return lv, what, r, r8, r0, r80, bits


def random_cases(n, seed):
    rnd = random.Random(seed)
    return [random_case(rnd) for _ in range(n)]

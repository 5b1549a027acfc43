#!/usr/bin/env python3
"""Quick GPU-vs-oracle parity sweep of the frame pipeline (development aid; the judged tests are tests/test_gpu_pipeline.py).
Usage: python tools/parity_mb.py [first_case [last_case]]   — prints the first mismatch of every case and keeps going."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402
from gpu_enc import GpuEncoder  # noqa: E402
from synth import synth_frames  # noqa: E402

FIELDS = ("type", "partition", "ref", "mv", "i16_mode", "i4_mode", "chroma_mode", "qp", "cbp_luma", "cbp_chroma", "transform8x8", "nnz", "cost", "aux")
CASES = [
    (64, 48, 3, dict(partitions=0)),
    (64, 48, 3, dict(partitions=0, subme=0, me_method=0, deblock=0)),
    (176, 144, 4, dict(partitions=0, subme=2)),
    (176, 144, 4, dict(partitions=0, subme=1)),
    (176, 144, 4, dict(partitions=0, subme=3)),
    (176, 144, 4, dict(partitions=0, subme=5)),
    (176, 144, 4, dict(partitions=2)),
    (176, 144, 4, dict(partitions=6, dct8x8=1)),
    (176, 144, 4, dict(partitions=1)),
    (176, 144, 4, dict(partitions=3, subme=5, chroma_me=1)),
    (176, 144, 6, dict(partitions=3, refs=3)),
    (176, 144, 6, dict(partitions=3, refs=3, mixed_refs=1)),
    (352, 288, 5, dict(partitions=7, refs=3, mixed_refs=1, dct8x8=1, chroma_me=1, subme=5, qp_i=26, qp_p=29)),
    (208, 120, 4, dict(partitions=7, dct8x8=1, fast_pskip=0, subme=4)),
    (208, 120, 4, dict(partitions=3, me_method=0, subme=2, refs=2)),
    (176, 144, 4, dict(partitions=7, dct8x8=1, aq_mode=1, refs=2, qp_i=24, qp_p=27)),
    (176, 144, 4, dict(partitions=3, subme=9, refs=2)),
    (96, 80, 5, dict(partitions=7, dct8x8=1, refs=4, mixed_refs=1, subme=5, chroma_me=1, qp_i=36, qp_p=40)),
    (176, 144, 8, dict(partitions=7, dct8x8=1, refs=5, mixed_refs=1, subme=5, chroma_me=1, qp_i=27, qp_p=30)),
    (352, 288, 4, dict(slices=4, partitions=7, dct8x8=1, refs=3, mixed_refs=1, subme=5, chroma_me=1)),
    (96, 80, 4, dict(partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0)),
    (176, 144, 4, dict(me_method=2)),
    (352, 288, 4, dict(me_method=2, partitions=3, refs=2, chroma_me=1, subme=5)),
    (208, 120, 4, dict(me_method=2, partitions=3, me_range=24, subme=5, mixed_refs=1, refs=3)),
    (720, 304, 3, dict(me_method=2, partitions=3, me_range=32, qp_i=30, qp_p=33)),
    (176, 144, 4, dict(me_method=3)),
    (208, 120, 4, dict(me_method=3, partitions=3, refs=2, me_range=8, chroma_me=1, subme=5)),
    (352, 288, 3, dict(me_method=2, partitions=7, dct8x8=1, refs=4, subme=9, chroma_me=1, mixed_refs=1, qp_i=26, qp_p=29)),
    (1280, 720, 3, dict(partitions=7, refs=3, mixed_refs=1, dct8x8=1, chroma_me=1, subme=5)),
    # RD mode decision (x264 subme 6 / 7, CAVLC bit counts): from the plainest candidate set up
    (64, 48, 3, dict(rd=1, subme=6, partitions=0)),
    (176, 144, 4, dict(rd=1, subme=6, partitions=2)),
    (176, 144, 4, dict(rd=1, subme=6, partitions=1)),
    (176, 144, 4, dict(rd=1, subme=7, partitions=3, refs=2, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
    (176, 144, 4, dict(rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
    (352, 288, 4, dict(rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2)),
    (96, 208, 4, dict(rd=1, subme=6, slices=3, partitions=7, dct8x8=1, refs=2, psy=1, psy_rd_q8=102, chroma_qp_offset=-1, aq_mode=1)),
    (96, 80, 4, dict(rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0)),
    # ... with CABAC: context states carried through the macroblock loop, candidates priced by the size-only coder
    (64, 48, 3, dict(cabac=1, rd=1, subme=6, partitions=0)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=2)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=1, refs=2)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=3, refs=3, mixed_refs=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=6, dct8x8=1)),
    (176, 144, 5, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
    (352, 288, 4, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2)),
    (96, 208, 4, dict(cabac=1, rd=1, subme=6, slices=3, partitions=7, dct8x8=1, refs=2, psy=1, psy_rd_q8=102, chroma_qp_offset=-1, aq_mode=1)),
    (96, 80, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0)),
    # trellis quantisation of the final encode, site by site (mask: 1 inter 4x4, 2 inter 8x8, 4 chroma, 8 I16x16, 16 I4x4, 32 I8x8; x264 --trellis 1 = 63)
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=1, trellis=1)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=1, trellis=4)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=5, dct8x8=1, trellis=2)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=2, trellis=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
    (96, 80, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, trellis=7)),
    (208, 120, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=40, qp_p=44, trellis=7)),
    (64, 48, 3, dict(cabac=1, rd=1, subme=6, partitions=0, trellis=8)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=2, trellis=16)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=4, dct8x8=1, trellis=32)),
    (176, 144, 5, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63)),
    (352, 288, 4, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, qp_i=30, qp_p=34, me_method=2, trellis=63)),
    (96, 208, 4, dict(cabac=1, rd=1, subme=6, slices=3, partitions=7, dct8x8=1, refs=2, aq_mode=1, trellis=63)),
    (96, 80, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0, trellis=63)),
    (208, 120, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=44, qp_p=47, trellis=63)),
    # --trellis 2 (bit 6): the search also in the intra analysis' block encodes and in every RD candidate
    (64, 48, 3, dict(cabac=1, rd=1, subme=6, partitions=0, trellis=127)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=6, partitions=2, trellis=127)),
    (176, 144, 4, dict(cabac=1, rd=1, subme=7, partitions=5, dct8x8=1, trellis=127)),
    (176, 144, 5, dict(cabac=1, rd=1, subme=7, partitions=7, dct8x8=1, refs=3, mixed_refs=1, chroma_me=1, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=127)),
    (96, 208, 4, dict(cabac=1, rd=1, subme=6, slices=3, partitions=7, dct8x8=1, refs=2, aq_mode=1, trellis=127)),
    (96, 80, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=8, qp_p=10, dct_decimate=0, fast_pskip=0, trellis=127)),
    (208, 120, 4, dict(cabac=1, rd=1, subme=6, partitions=7, dct8x8=1, qp_i=44, qp_p=47, me_method=2, trellis=127)),
]


def first_diff(g_mb, o_mb, g_lv, o_lv, g_rec, o_rec, mbw):
    for f in FIELDS:
        a, b = g_mb[f], o_mb[f]
        if not np.array_equal(a, b):
            bad = np.nonzero((a != b).reshape(len(a), -1).any(1))[0]
            i = int(bad[0])
            return (f"field {f}: {len(bad)} MBs differ; first MB {i} (x={i % mbw}, y={i // mbw}) gpu={a[i]} oracle={b[i]} | types g/o {g_mb['type'][i]}/{o_mb['type'][i]} "
                    f"part {g_mb['partition'][i]}/{o_mb['partition'][i]} mv {g_mb['mv'][i].tolist()}/{o_mb['mv'][i].tolist()} ref {g_mb['ref'][i].tolist()}/{o_mb['ref'][i].tolist()} "
                    f"cost {g_mb['cost'][i]}/{o_mb['cost'][i]} aux {g_mb['aux'][i].tolist()}/{o_mb['aux'][i].tolist()}")
    if not np.array_equal(g_lv, o_lv):
        bad = np.nonzero((g_lv != o_lv).any(1))[0]
        i = int(bad[0])
        pos = np.nonzero(g_lv[i] != o_lv[i])[0]
        return f"levels: {len(bad)} MBs differ; first MB {i} (x={i % mbw}, y={i // mbw}, type {o_mb['type'][i]}) idx {pos[:8]} gpu={g_lv[i][pos[:8]]} oracle={o_lv[i][pos[:8]]}"
    if not np.array_equal(g_rec, o_rec):
        pos = np.nonzero(g_rec != o_rec)[0]
        return f"recon: {len(pos)} bytes differ, first offsets {pos[:8]}"
    return None


def main():
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else len(CASES) - 1
    nbad = 0
    for ci in range(lo, hi + 1):
        w, h, n, kw = CASES[ci]
        frames = synth_frames(w, h, n, seed=w * 7 + h)
        cfg = O.default_config(w, h, **kw)
        og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
        msg = None
        for i, f in enumerate(frames):
            st = 2 if i == 0 else 0
            o_mb, o_lv = og.encode(f, st)
            g_mb, g_lv = gg.encode([f], st)
            d = first_diff(g_mb[0], o_mb, g_lv[0], o_lv, gg.recon(0), og.recon(), (w + 15) // 16)
            if d:
                msg = f"frame {i}: {d}"
                break
        print(f"case {ci} {w}x{h} {kw}: {'OK' if not msg else 'MISMATCH ' + msg}", flush=True)
        nbad += msg is not None
        og.close(); gg.close()
    print(f"{nbad} of {hi - lo + 1} cases differ")


if __name__ == "__main__":
    main()

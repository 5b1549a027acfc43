#!/usr/bin/env python3
"""Regenerates tests/golden/*.json|npz from the CPU oracle (oracle/liboracle.so).

The reference tree has no tests, fixtures or golden vectors for this path and its
implementation (libx264) is not in /root/reference, so nothing can be captured from the reference itself:
these fixtures pin the oracle (which tests/test_oracle_spec.py pins to ITU-T H.264) so that a later edit of
either the oracle or the HIP path that changes results is caught on both the CPU and the GPU box.
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
from synth import synth_frames  # noqa: E402

CASES = [("p176x144", 176, 144, 5, {}), ("p208x120_q30", 208, 120, 3, dict(qp_i=27, qp_p=30)),
         ("p64x48_nodeblock", 64, 48, 3, dict(deblock=0)),
         ("p176x144_medium", 176, 144, 5, dict(refs=3, partitions=7, dct8x8=1)),
         ("p208x120_i8x8_only", 208, 120, 3, dict(partitions=4, dct8x8=1, qp_i=24, qp_p=27)),
         ("p176x144_medium_chroma_me", 176, 144, 5, dict(refs=3, partitions=7, dct8x8=1, chroma_me=1)),
         ("p176x144_lowqp_umh", 176, 144, 4, dict(qp_i=10, qp_p=13, refs=2, partitions=7, dct8x8=1, chroma_me=1, me_method=2)),
         ("p176x144_x264_medium_me", 176, 144, 6, dict(refs=3, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1)),
         ("p176x144_aq", 176, 144, 4, dict(refs=2, partitions=7, dct8x8=1, chroma_me=1, aq_mode=1, aq_strength=1.0397, qp_i=23, qp_p=26)),
         # round 4: the headline toolset — RD mode decision priced with CABAC sizes (subme 7), psy-rd, trellis 1 / 2 — and RD refinement (subme 8)
         ("p176x144_medium_rd_cabac_trellis", 176, 144, 5, dict(refs=3, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256,
                                                               chroma_qp_offset=-2, trellis=63)),
         ("p176x144_rd_cavlc", 176, 144, 4, dict(refs=2, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2)),
         ("p128x96_trellis2_umh", 128, 96, 4, dict(refs=2, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256,
                                                   chroma_qp_offset=-2, trellis=127, me_method=2)),
         ("p176x144_subme8_rd_refine", 176, 144, 4, dict(refs=3, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=63, subme=8, psy=1, psy_rd_q8=256,
                                                        chroma_qp_offset=-2, trellis=127))]
MEDIUM_B = dict(refs=3, dpb=4, weightb=1, partitions=7, dct8x8=1, chroma_me=1, mixed_refs=1, cabac=1, rd=1, subme=7, psy=1, psy_rd_q8=256, chroma_qp_offset=-2, trellis=63)
# (name, w, h, display-order types, seed, weightp, config overrides): mini-GOPs through the product's DPB model (host/dpb.hpp) and oracle encode_pic
B_CASES = [("b176x144_medium_weightp2", 176, 144, "IBBBPBBBP", 5, 2, {}), ("b128x96_ref5_umh", 128, 96, "IBBPBP", 7, 0, dict(refs=5, dpb=5, me_method=2)),
           # round 4's paths: x264 --subme 9 (RD refinement in B slices, deblock-aware RD), B analysis without RD, RD on CAVLC counts, temporal direct / direct auto
           ("b176x144_subme9_refine", 176, 144, "IBBBPBBP", 33, 0, dict(subme=9, rd=63 | 64)),
           ("b176x144_subme5_without_rd", 176, 144, "IBBBPBBP", 21, 0, dict(rd=0, trellis=0, subme=5, psy_rd_q8=0)),
           ("b176x144_rd_on_cavlc_counts", 176, 144, "IBBBPBBP", 51, 0, dict(cabac=0, trellis=0)),
           ("b128x96_direct_temporal", 128, 96, "IBBBPBBBP", 11, 0, dict(_direct="temporal")),
           ("b128x96_direct_auto_subme8", 128, 96, "IBBBPBBBP", 11, 0, dict(_direct="auto", subme=8, rd=63))]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pipeline_case(w, h, n, kw):
    enc = O.OracleEncoder(O.default_config(w, h, **kw))
    out = []
    for i, f in enumerate(synth_frames(w, h, n, seed=w * 7 + h)):
        mbs, lv = enc.encode(f, 2 if i == 0 else 0)
        out.append({"mb": sha(mbs.view(np.uint8)), "levels": sha(lv), "recon": sha(enc.recon()),
                    "types": np.bincount(mbs["type"], minlength=7).tolist(), "nonzero_levels": int((lv != 0).sum())})
    enc.close()
    return out


def b_case(w, h, types, seed, weightp, over):
    os.environ.setdefault("X264_HOST_STUB", "1")          # the DPB model is host code: linked against the stand-in device library on a CPU-only box
    import bgop
    import host_lib as HL
    kw = dict(MEDIUM_B, **over)
    direct = kw.pop("_direct", "spatial")
    cfg = O.default_config(w, h, **kw)
    enc = O.OracleEncoder(cfg)
    pics = []
    stream, recons, order, pocs = bgop.encode_gop(HL, enc, synth_frames(w, h, len(types), seed=seed), types, cfg, 20, 23, 25, kw["refs"], 3, 1, weightp, pics, None, direct)
    return {"stream": sha(np.frombuffer(bytes(stream), dtype=np.uint8)), "bytes": len(stream), "order": [list(o) for o in order],
            "per_picture": [{"mb": sha(m.view(np.uint8)), "recon": sha(r), "types": np.bincount(m["type"], minlength=11).tolist()}
                            for (_, m), r in zip([e for e in pics if len(e) == 2], recons)]}          # (--direct auto also logs its mode choices there)


def slicetype_case():
    """frame costs of (p0, p1, b) triples on the half-resolution planes (oracle/slicetype.c): what the slice-type decisions and the macroblock-tree read"""
    w, h = 176, 144
    fr = synth_frames(w, h, 5, seed=33)
    st = O.OracleSlicetype(w, h, slots=8, bframes=3)
    for i, f in enumerate(fr):
        st.put(i, f)
    out = {"w": w, "h": h, "seed": 33, "costs": {}}
    for (s0, s1, sb) in [(0, 0, 0), (0, 1, 1), (0, 2, 2), (0, 2, 1), (0, 4, 2), (1, 4, 3), (0, 3, 3)]:
        out["costs"][f"{s0},{s1},{sb}"] = int(st.cost(s0, s1, sb, sb - s0, s1 - sb))
    out["intra_costs_sha"] = sha(st.intra_costs(2)); out["mvs_sha"] = sha(st.mvs(2, 0, 2))
    st.close()
    return out


def prim_vectors():
    rng = np.random.default_rng(0x264)
    a = rng.integers(0, 256, (32, 16, 16), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    d = {"a": a, "b": b}
    for m in ("sad", "satd", "sa8d", "ssd"):
        d[m + "_16x16"] = O.metric(m, a, b)
    a8, b8 = np.ascontiguousarray(a[:, :8, :8]), np.ascontiguousarray(b[:, :8, :8])
    for m in ("sad", "satd", "sa8d"):
        d[m + "_8x8"] = O.metric(m, a8, b8)
    enc = np.ascontiguousarray(a[:, :4, :4])
    pred = np.ascontiguousarray(b[:, 4:8, 4:8])
    for qp in (0, 23, 37, 51):
        for lst in (0, 1):
            c, l, r = O.dctq4x4(enc, pred, qp, lst)
            d[f"coef_q{qp}_l{lst}"], d[f"lev_q{qp}_l{lst}"], d[f"rec_q{qp}_l{lst}"] = c, l, r
    d["enc4"], d["pred4"] = enc, pred
    return d


def csp_vectors():
    """hashes of the colourspace-ingest oracle (oracle/csp.c) on seeded frames, every format x flip x matrix/range"""
    out = {}
    rng = np.random.default_rng(0xC59)
    w, h = 48, 20
    for name in ("I420", "YV12", "YV16", "YV24", "YUYV", "UYVY", "BGR", "BGRA"):
        for flip in (0, 1):
            csp = O.CSP[name] | (O.CSP["VFLIP"] if flip else 0)
            buf = rng.integers(0, 256, O.csp_img_fill(csp, w, h)[0], dtype=np.uint8)
            for mat, full in ([(0, 0)] if name not in ("BGR", "BGRA") else [(0, 0), (0, 1), (1, 0), (1, 1)]):
                out[f"{name}_flip{flip}_m{mat}_r{full}"] = {"in": sha(buf), "out": sha(O.csp_to_i420(buf, csp, w, h, mat, full))}
    return {"w": w, "h": h, "seed": 0xC59, "cases": out}


if __name__ == "__main__":
    json.dump(csp_vectors(), open(os.path.join(HERE, "oracle_csp.json"), "w"), indent=1)
    js = {name: {"w": w, "h": h, "frames": n, "cfg": kw, "per_frame": pipeline_case(w, h, n, kw)} for name, w, h, n, kw in CASES}
    json.dump(js, open(os.path.join(HERE, "oracle_pipeline.json"), "w"), indent=1)
    jb = {name: dict(w=w, h=h, types=t, seed=sd, weightp=wp, cfg=ov, **b_case(w, h, t, sd, wp, ov)) for name, w, h, t, sd, wp, ov in B_CASES}
    json.dump(jb, open(os.path.join(HERE, "oracle_bframes.json"), "w"), indent=1)
    json.dump(slicetype_case(), open(os.path.join(HERE, "oracle_slicetype.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "prim_vectors.npz"), **prim_vectors())
    print("golden fixtures written")

"""CPU: the oracle against the committed golden fixtures (tests/golden, made by make_golden.py), sanity
of the oracle encoder, and the C-ABI export surface of libx264gpu.so (no compute calls without a GPU)."""
import hashlib
import json
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from synth import psnr, synth_frames

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_prim_golden_vectors():
    g = np.load(os.path.join(GOLD, "prim_vectors.npz"))
    a, b = g["a"], g["b"]
    for m in ("sad", "satd", "sa8d", "ssd"):
        np.testing.assert_array_equal(O.metric(m, a, b), g[m + "_16x16"])
    a8, b8 = np.ascontiguousarray(a[:, :8, :8]), np.ascontiguousarray(b[:, :8, :8])
    for m in ("sad", "satd", "sa8d"):
        np.testing.assert_array_equal(O.metric(m, a8, b8), g[m + "_8x8"])
    for qp in (0, 23, 37, 51):
        for lst in (0, 1):
            c, l, r = O.dctq4x4(g["enc4"], g["pred4"], qp, lst)
            np.testing.assert_array_equal(c, g[f"coef_q{qp}_l{lst}"])
            np.testing.assert_array_equal(l, g[f"lev_q{qp}_l{lst}"])
            np.testing.assert_array_equal(r, g[f"rec_q{qp}_l{lst}"])


@pytest.mark.parametrize("case", ["p176x144", "p208x120_q30", "p64x48_nodeblock", "p176x144_medium", "p208x120_i8x8_only", "p176x144_medium_chroma_me", "p176x144_lowqp_umh", "p176x144_x264_medium_me", "p176x144_aq",
                                  "p176x144_medium_rd_cabac_trellis", "p176x144_rd_cavlc", "p128x96_trellis2_umh", "p176x144_subme8_rd_refine"])
def test_pipeline_golden(case):
    js = json.load(open(os.path.join(GOLD, "oracle_pipeline.json")))[case]
    w, h = js["w"], js["h"]
    enc = O.OracleEncoder(O.default_config(w, h, **js["cfg"]))
    for i, (f, exp) in enumerate(zip(synth_frames(w, h, js["frames"], seed=w * 7 + h), js["per_frame"])):
        mbs, lv = enc.encode(f, 2 if i == 0 else 0)
        assert sha(mbs.view(np.uint8)) == exp["mb"], f"frame {i} records"
        assert sha(lv) == exp["levels"], f"frame {i} levels"
        assert sha(enc.recon()) == exp["recon"], f"frame {i} recon"


@pytest.mark.parametrize("case", ["b176x144_medium_weightp2", "b128x96_ref5_umh", "b176x144_subme9_refine", "b176x144_subme5_without_rd", "b176x144_rd_on_cavlc_counts",
                                  "b128x96_direct_temporal", "b128x96_direct_auto_subme8"])
def test_bframes_golden(case):
    """B mini-GOPs of the headline toolset (spatial direct, two-list searches, implicit weights, B RD decision, --weightp 2's duplicate) through the
    product's DPB model and host CABAC writer: records, reconstruction and the written stream against the committed hashes"""
    import subprocess
    import sys
    # (the DPB model is host code linked against the stand-in device library: its own process, as in test_bframes_cpu.py's sessions)
    code = ("import os, sys, json; os.environ['X264_HOST_STUB'] = '1'; sys.path.insert(0, %r); sys.path.insert(0, %r);"
            "import make_golden as G; js = json.load(open(os.path.join(%r, 'oracle_bframes.json')))[%r];"
            "r = G.b_case(js['w'], js['h'], js['types'], js['seed'], js['weightp'], js['cfg']);"
            "assert r['per_picture'] == js['per_picture'], 'records / reconstruction'; assert r['stream'] == js['stream'] and r['bytes'] == js['bytes'], 'stream'"
            % (os.path.dirname(__file__), GOLD, GOLD, case))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]


def test_slicetype_golden():
    """the lookahead's frame costs (oracle/slicetype.c) against the committed values"""
    sys_path = os.path.join(GOLD)
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(sys_path, "make_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    assert G.slicetype_case() == json.load(open(os.path.join(GOLD, "oracle_slicetype.json")))


def test_csp_golden():
    """oracle/csp.c against the committed hashes (tests/golden/oracle_csp.json)"""
    js = json.load(open(os.path.join(GOLD, "oracle_csp.json")))
    rng = np.random.default_rng(js["seed"])
    w, h = js["w"], js["h"]
    for name in ("I420", "YV12", "YV16", "YV24", "YUYV", "UYVY", "BGR", "BGRA"):
        for flip in (0, 1):
            csp = O.CSP[name] | (O.CSP["VFLIP"] if flip else 0)
            buf = rng.integers(0, 256, O.csp_img_fill(csp, w, h)[0], dtype=np.uint8)
            for mat, full in ([(0, 0)] if name not in ("BGR", "BGRA") else [(0, 0), (0, 1), (1, 0), (1, 1)]):
                exp = js["cases"][f"{name}_flip{flip}_m{mat}_r{full}"]
                assert sha(buf) == exp["in"] and sha(O.csp_to_i420(buf, csp, w, h, mat, full)) == exp["out"], (name, flip, mat, full)


def test_oracle_encoder_quality_and_monotonic_qp():
    w, h = 176, 144
    frames = synth_frames(w, h, 4, seed=3)
    res = {}
    for qp in (18, 28, 38):
        enc = O.OracleEncoder(O.default_config(w, h, qp_i=qp - 3, qp_p=qp))
        ps, nz = [], 0
        for i, f in enumerate(frames):
            mbs, lv = enc.encode(f, 2 if i == 0 else 0)
            ps.append(psnr(enc.recon()[:w * h], f[:w * h]))
            nz += int((lv != 0).sum())
        res[qp] = (min(ps), nz)
    assert res[18][0] > res[28][0] > res[38][0] > 24.0
    assert res[18][1] > res[28][1] > res[38][1]


def test_oracle_static_scene_goes_inter_zero_mv():
    """edge case: identical frames -> P frame is all inter with zero motion and no residual"""
    w, h = 96, 64
    f = synth_frames(w, h, 1, seed=5)[0]
    enc = O.OracleEncoder(O.default_config(w, h, qp_i=20, qp_p=30))
    enc.encode(f, 2)
    rec_i = enc.recon()
    mbs, lv = enc.encode(rec_i, 0)       # feed the reconstruction back: perfectly predictable
    assert (mbs["type"] == 6).all() and (mbs["mv"] == 0).all() and not lv.any()      # P_Skip, found by the analysis itself


def test_abi_exports_every_declared_symbol():
    """include/x264gpu.h is the contract: every function it declares is exported by libx264gpu.so"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "x264gpu.h")).read()
    declared = set(re.findall(r"\b(x264gpu_[a-z0-9_]+)\s*\(", hdr))
    from x264vfw_amd import lib
    assert declared, "no declarations parsed"
    missing = sorted(d for d in declared if d not in lib.EXPORTS)
    assert not missing, f"declared in x264gpu.h but not bound: {missing}"
    import ctypes
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), f"{name} not exported by libx264gpu.so"
    assert lib.x264gpu_abi_version() == 2


def test_no_gpu_means_loud_failure():
    """without a device the product reports it; there is no CPU fallback to fall into"""
    import torch
    from x264vfw_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.x264gpu_device_count() == 0
    import ctypes as C
    h = C.c_void_p()
    cfg = O.default_config(64, 48)
    assert lib.x264gpu_encoder_create(C.byref(h), C.byref(cfg)) != 0


def test_pred8_table_comes_from_the_standards_equations():
    """x264vfw_amd/csrc/pred8_table.inc (the product's Intra_8x8 lookup) is derived from tests/spec_ref.py's per-pixel equations of 8.3.2.2,
    not from the oracle: regenerating it gives the committed file"""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_pred8_table", os.path.join(root, "tools", "gen_pred8_table.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    assert g.text(g.table()) == open(g.PATH).read()


def test_fixed_point_tables_of_the_product_match_libm():
    """the AQ / macroblock-tree tables the product carries as literals (include/x264gpu_log2f_lut.inc, x264gpu_exp2_lut.inc) against the oracle's own
    derivation with libm (oracle/fixlut.h) and against numpy: the oracle does not read the product's files.  x264_log2_lut is written in x264 as
    five-decimal literals (0.00000, 0.01123, 0.02237 ... 0.99435): a double literal stored in a float — the first row and the last entry from memory of
    [x264-upstream] common/tables.c"""
    import ctypes as C
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in os.listdir(os.path.join(root, "oracle")):
        if f.endswith((".c", ".cpp", ".h", ".hpp")):
            assert "_lut.inc" not in open(os.path.join(root, "oracle", f)).read().replace("x264gpu_log2f_lut.inc, include/x264gpu_exp2_lut.inc", ""), f
    text = lambda name: re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", name)).read(), flags=re.S)
    lg, ex = np.zeros(128, np.float32), np.zeros(64, np.uint16)
    O.L.x264o_fixed_point_luts.argtypes = [C.c_void_p, C.c_void_p]
    O.L.x264o_fixed_point_luts(lg.ctypes.data, ex.ctypes.data)
    lit = [np.float32(float(x)) for x in re.findall(r"(\d\.\d{5})f", text("x264gpu_log2f_lut.inc"))]
    assert len(lit) == 128 and np.array(lit, np.float32).tobytes() == lg.tobytes()
    assert lg.tobytes() == np.array([float("%.5f" % v) for v in np.log2(1 + np.arange(128) / 128)], np.float32).tobytes()
    assert ["%.5f" % v for v in lg[:8]] == ["0.00000", "0.01123", "0.02237", "0.03342", "0.04439", "0.05528", "0.06609", "0.07682"] and "%.5f" % lg[127] == "0.99435"
    assert [int(x) for x in re.findall(r"-?\d+", text("x264gpu_exp2_lut.inc"))] == ex.tolist() == np.rint(256 * (2 ** (np.arange(64) / 64) - 1)).astype(int).tolist()


def test_float_primitives_of_aq_and_mbtree_known_answers():
    """x264_log2 / x264_exp2fix8 / x264_ratecontrol_mb_qp as the oracle restates them (oracle/fixlut.h), against values worked out by hand from x264's
    definitions and against the same expressions in numpy single floats"""
    import ctypes as C
    L = O.L
    L.x264o_log2_f.restype = C.c_float; L.x264o_log2_f.argtypes = [C.c_uint32]
    L.x264o_exp2fix8_f.restype = C.c_int; L.x264o_exp2fix8_f.argtypes = [C.c_float]
    L.x264o_mb_qp_f.restype = C.c_int; L.x264o_mb_qp_f.argtypes = [C.c_float, C.c_float]
    f32 = np.float32
    # x264_log2: exact on powers of two, table + exponent elsewhere (the argument's top 7 bits below the leading one)
    assert [L.x264o_log2_f(1 << k) for k in (0, 1, 10, 31)] == [0.0, 1.0, 10.0, 31.0]
    assert L.x264o_log2_f(3) == f32(0.58496) + f32(1) and L.x264o_log2_f(0xffffffff) == f32(0.99435) + f32(31)
    assert L.x264o_log2_f(22000) == f32(float("%.5f" % np.log2(1 + ((22000 << 17 >> 24) & 0x7f) / 128))) + f32(14)          # 22000 = 2^14 * 1.3427...
    # x264_exp2fix8: 256 at 0, halves every 6 quantiser steps, saturates
    assert [L.x264o_exp2fix8_f(x) for x in (0.0, 6.0, -6.0, 12.0, 3.0, 49.0, -48.1)] == [256, 128, 512, 64, 181, 0, 0xffff]
    rnd = np.random.default_rng(1)
    for x in rnd.uniform(-20, 20, 200).astype(np.float32):
        i = int(f32(f32(x * f32(-64.0 / 6.0)) + f32(512.5)))
        want = 0 if i < 0 else 0xffff if i > 1023 else ((int(np.rint(256 * (2 ** ((i & 63) / 64) - 1))) + 256) << (i >> 6)) >> 8
        assert L.x264o_exp2fix8_f(float(x)) == want, x
    # x264_ratecontrol_mb_qp: (int)(qpm + offset + 0.5f), two float additions
    assert [L.x264o_mb_qp_f(28.4, o) for o in (0.0, 0.09, 0.11, -0.9, -0.91, 3.1)] == [28, 28, 29, 28, 27, 32]
    for qpm, off in zip(rnd.uniform(10, 51, 200).astype(np.float32), rnd.uniform(-8, 8, 200).astype(np.float32)):
        assert L.x264o_mb_qp_f(float(qpm), float(off)) == int(f32(f32(qpm + off) + f32(0.5)))


def test_mbtree_float_arithmetic_against_numpy():
    """oracle/slicetype.c's macroblock-tree (mbtree_propagate_cost / mbtree_propagate_list / macroblock_tree_finish in x264's single floats) against the same expressions in numpy float32,
    block by block, on the lookahead's own costs and vectors of a small clip: propagate amounts, the bilinear split with saturation, the finished offsets with a weightdelta"""
    w, h, n = 176, 144, 5
    frames = synth_frames(w, h, n, seed=31)
    st = O.OracleSlicetype(w, h, slots=8, bframes=3, do_edges=1)
    bw, bh = (w // 2 + 7) // 8, (h // 2 + 7) // 8
    nb = bw * bh
    aqs = []
    for i, f in enumerate(frames):
        st.put(i, f)
        aqs.append(O.aq_offsets(f, w, h))
        st.set_aq(i, aqs[-1])
    f32 = np.float32
    lut = np.array([float("%.5f" % v) for v in np.log2(1 + np.arange(128) / 128)], f32)
    exp2 = np.rint(256 * (2 ** (np.arange(64) / 64) - 1)).astype(int)

    def log2(x):
        lz = 32 - int(x).bit_length()
        return f32(lut[((int(x) << lz) >> 24) & 0x7f] + f32(31 - lz))

    def exp2fix8(x):
        i = int(f32(f32(f32(x) * f32(-64.0 / 6.0)) + f32(512.5)))
        return 0 if i < 0 else 0xffff if i > 1023 else ((int(exp2[i & 63]) + 256) << (i >> 6)) >> 8

    # the walk x264's macroblock_tree makes over I P B P P with the B picture between pictures 1 and 3: (p0, p1, b, referenced) in its order
    st.cost(0, 0, 0, 0, 0)
    steps = [(3, 4, 4, 1), (1, 3, 2, 0), (1, 3, 3, 1), (0, 1, 1, 1)]
    prop = np.zeros((n, nb), np.int64)
    for p0, p1, b, referenced in steps:
        d0, d1 = b - p0, p1 - b
        st.cost(p0, p1, b, d0, d1)
        st.propagate(p0, p1, b, d0, d1, referenced)
        intra = np.minimum(st.intra_costs(b), 16383)
        lc = st.lowres_costs(b, d0, d1)
        dsf = ((d0 << 8) + ((d0 + d1) >> 1)) // (d0 + d1) if d1 > 0 else 256
        bipw = 64 - (dsf >> 2) if d1 > 0 else 32
        mv = [st.mvs(b, 0, d0), st.mvs(b, 1, d1) if d1 > 0 else None]
        refs = [p0, p1]
        for i in range(nb):
            ic, best, used = int(intra[i]), int(lc[i]) & 16383, int(lc[i]) >> 14
            inter = min(best, ic)
            amount = 0
            if ic:
                pin = f32(min(int(prop[b][i]), 32767) if referenced else 0)
                amt = f32(pin + f32(f32(ic * exp2fix8(aqs[b][i])) * f32(1.0 / 512.0)))
                amount = min(int(f32(f32(f32(amt * f32(ic - inter)) / f32(ic)) + f32(0.5))), 32767)
            for l in range(2 if d1 > 0 else 1):
                if not used & (1 << l):
                    continue
                la = (amount * (bipw if l == 0 else 64 - bipw) + 32) >> 6 if used == 3 else amount
                x, y = int(mv[l][i][0]), int(mv[l][i][1])
                tgt = prop[refs[l]]
                bx, by = i % bw, i // bw

                def add(xx, yy, v):
                    if 0 <= xx < bw and 0 <= yy < bh:
                        tgt[yy * bw + xx] = min(tgt[yy * bw + xx] + v, 32767)
                if not (x | y):
                    add(bx, by, la)
                    continue
                mbx, mby, x, y = (x >> 5) + bx, (y >> 5) + by, x & 31, y & 31
                add(mbx, mby, ((32 - y) * (32 - x) * la + 512) >> 10); add(mbx + 1, mby, ((32 - y) * x * la + 512) >> 10)
                add(mbx, mby + 1, (y * (32 - x) * la + 512) >> 10); add(mbx + 1, mby + 1, (y * x * la + 512) >> 10)
        for k in range(n):
            assert np.array_equal(st.propagate_cost(k), np.minimum(prop[k], 32767)), (p0, p1, b, k)
    assert prop[0].sum() > 0 and prop[1].sum() > 0
    for slot, strength, wd in ((1, 2.0, 0.0), (0, 1.5, 0.125)):
        got = st.finish(slot, strength, wd)
        intra = np.minimum(st.intra_costs(slot), 16383)
        want = np.zeros(nb, f32)
        for i in range(nb):
            a = f32(aqs[slot][i])
            ic = (int(intra[i]) * exp2fix8(a) + 128) >> 8
            want[i] = a
            if ic:
                ratio = f32(f32(log2(ic + 2 * min(int(prop[slot][i]), 32767)) - log2(ic)) + f32(wd))
                want[i] = f32(a - f32(f32(strength) * ratio))
        assert got.tobytes() == want.tobytes(), (slot, np.nonzero(got != want)[0][:5])
    st.close()


def test_aq_mode_1_offsets_against_numpy():
    """x264_adaptive_quant_frame, mode 1, as the oracle restates it (single floats) against numpy: AC energy of every 16x16 luma + two 8x8 chroma blocks, strength x (x264_log2(energy) - 14.427f)"""
    w, h = 176, 144
    f = synth_frames(w, h, 1, seed=7)[0]
    f32 = np.float32
    lut = np.array([float("%.5f" % v) for v in np.log2(1 + np.arange(128) / 128)], f32)
    Y = f[:w * h].reshape(h, w).astype(np.int64)
    U, V = f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2).astype(np.int64), f[w * h * 5 // 4:].reshape(h // 2, w // 2).astype(np.int64)
    strength = f32(f32(0.8) * f32(1.0397))
    got = O.aq_offsets(f, w, h, float(strength))
    want = np.zeros_like(got)
    for by in range(h // 16):
        for bx in range(w // 16):
            y, u, v = Y[by * 16:by * 16 + 16, bx * 16:bx * 16 + 16], U[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8], V[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]
            var = lambda b_, sh: int((b_ * b_).sum()) - ((int(b_.sum()) ** 2) >> sh)
            e = max((var(y, 8) + var(u, 6) + var(v, 6)) & 0xffffffff, 1)
            lz = 32 - e.bit_length()
            lg = f32(lut[((e << lz) >> 24) & 0x7f] + f32(31 - lz))
            want[by * (w // 16) + bx] = f32(strength * f32(lg - f32(14.427)))
    assert got.tobytes() == want.tobytes() and np.abs(got).max() > 0.5


@pytest.mark.parametrize("mode", [2, 3])
def test_aq_modes_2_and_3_against_numpy(mode):
    """x264_adaptive_quant_frame, modes 2 / 3 (auto-variance, with the bias to dark scenes): (energy + 1)^(1/8) per macroblock — three IEEE square roots here —, the picture's mean and
    mean square summed in raster order in single floats, strength x (value - average) [+ strength x (1 - 14 / value^2)]: the oracle against numpy float32, bit for bit"""
    w, h = 176, 144
    f = synth_frames(w, h, 1, seed=11)[0]
    f32 = np.float32
    Y = f[:w * h].reshape(h, w).astype(np.int64)
    U, V = f[w * h:w * h * 5 // 4].reshape(h // 2, w // 2).astype(np.int64), f[w * h * 5 // 4:].reshape(h // 2, w // 2).astype(np.int64)
    aqs = f32(1.2)
    got = O.aq_offsets_mode(f, w, h, mode, float(aqs))
    adj, s1, s2 = [], f32(0), f32(0)
    var = lambda b_, sh: int((b_ * b_).sum()) - ((int(b_.sum()) ** 2) >> sh)
    for by in range(h // 16):
        for bx in range(w // 16):
            y, u, v = Y[by * 16:by * 16 + 16, bx * 16:bx * 16 + 16], U[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8], V[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]
            e = (var(y, 8) + var(u, 6) + var(v, 6)) & 0xffffffff
            q = np.sqrt(np.sqrt(np.sqrt(f32(f32(e) + f32(1)))))
            adj.append(q); s1 = f32(s1 + q); s2 = f32(s2 + f32(q * q))
    n = f32(len(adj))
    mean, msq = f32(s1 / n), f32(s2 / n)
    strength = f32(aqs * mean)
    avg = f32(mean - f32(f32(f32(0.5) * f32(msq - f32(14))) / mean))
    want = np.zeros(len(adj), f32)
    for i, q in enumerate(adj):
        o = f32(strength * f32(q - avg))
        if mode == 3:
            o = f32(o + f32(aqs * f32(f32(1) - f32(f32(14) / f32(q * q)))))
        want[i] = o
    assert got.tobytes() == want.tobytes() and np.abs(got).max() > 0.2


def test_ip_stream_mbtree_against_numpy():
    """oracle/lookahead.c x264o_mbtree (the macroblock-tree of an I / P stream: every picture hands its explained cost to its predecessor) against numpy float32, bit for bit"""
    w, h, n = 176, 144, 5
    frames = synth_frames(w, h, n, seed=23)
    ol = O.OracleLookahead(w, h)
    infos = [ol.frame_cost(f, i == 0)[1] for i, f in enumerate(frames)]
    aqs = [O.aq_offsets(f, w, h) for f in frames]
    bw, bh = (w + 15) // 16, (h + 15) // 16
    nb = bw * bh
    f32 = np.float32
    lut = np.array([float("%.5f" % v) for v in np.log2(1 + np.arange(128) / 128)], f32)
    exp2 = np.rint(256 * (2 ** (np.arange(64) / 64) - 1)).astype(int)
    log2 = lambda x: f32(lut[((int(x) << (32 - int(x).bit_length())) >> 24) & 0x7f] + f32(int(x).bit_length() - 1))

    def exp2fix8(x):
        i = int(f32(f32(f32(x) * f32(-64.0 / 6.0)) + f32(512.5)))
        return 0 if i < 0 else 0xffff if i > 1023 else ((int(exp2[i & 63]) + 256) << (i >> 6)) >> 8
    strength = f32(2.0)
    got = O.mbtree(bw, bh, infos, aqs, float(strength))
    prop = np.zeros((n, nb), np.int64)
    for j in range(n - 1, 0, -1):
        fi, ref = infos[j], prop[j - 1]
        for i in range(nb):
            intra, best = min(int(fi[i][0]), 16383), min(int(fi[i][1]), 16383)
            inter = min(best, intra)
            amount = 0
            if intra:
                amt = f32(f32(int(prop[j][i])) + f32(f32(intra * exp2fix8(aqs[j][i])) * f32(1.0 / 512.0)))
                amount = min(int(f32(f32(f32(amt * f32(intra - inter)) / f32(intra)) + f32(0.5))), 32767)
            if not fi[i][3]:
                continue
            v = int(fi[i][2])
            x, y = ((v & 0xffff) ^ 0x8000) - 0x8000, v >> 16
            bx, by = i % bw, i // bw

            def add(xx, yy, val):
                if 0 <= xx < bw and 0 <= yy < bh:
                    ref[yy * bw + xx] = min(ref[yy * bw + xx] + val, 32767)
            if not (x | y):
                add(bx, by, amount)
                continue
            mbx, mby, x, y = (x >> 5) + bx, (y >> 5) + by, x & 31, y & 31
            add(mbx, mby, ((32 - y) * (32 - x) * amount + 512) >> 10); add(mbx + 1, mby, ((32 - y) * x * amount + 512) >> 10)
            add(mbx, mby + 1, (y * (32 - x) * amount + 512) >> 10); add(mbx + 1, mby + 1, (y * x * amount + 512) >> 10)
    want = np.zeros(nb, f32)
    for i in range(nb):
        a = f32(aqs[0][i])
        ic = (min(int(infos[0][i][0]), 16383) * exp2fix8(a) + 128) >> 8
        want[i] = a if not ic else f32(a - f32(strength * f32(f32(log2(ic + 2 * int(prop[0][i])) - log2(ic)) + f32(0))))
    assert got.tobytes() == want.tobytes(), np.nonzero(got != want)[0][:5]
    assert (got < aqs[0]).sum() > nb // 4

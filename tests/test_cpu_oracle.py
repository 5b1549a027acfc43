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

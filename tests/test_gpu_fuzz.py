"""-m gpu: randomised parity sweep — HIP frame pipeline == oracle (records, levels, reconstruction) over random picture sizes
(including non-multiples of 16), quantisers 0..51, deblock / chroma-qp offsets and every toolset combination the config struct
can express (refs 1..5, partitions, 8x8 transform + Intra_8x8, subme 0..9, me dia/hex, merange, decimate), with a second
IDR inside some sequences.  Seeds are fixed so a failure names a reproducible case."""
import random

import numpy as np
import pytest

import oracle_lib as O
from synth import synth_frames

pytestmark = pytest.mark.gpu


def random_case(rnd):
    w = max(16, 16 * rnd.randint(1, 14) - rnd.choice([0, 0, 2, 6, 14]))
    h = max(16, 16 * rnd.randint(1, 10) - rnd.choice([0, 0, 2, 8, 12]))
    dct = rnd.randint(0, 1)
    kw = dict(refs=rnd.randint(1, 5), partitions=rnd.choice([0, 1, 2, 3, 4, 5, 6, 7]) if dct else rnd.choice([0, 1, 2, 3]), dct8x8=dct,
              subme=rnd.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9]), me_method=rnd.choice([0, 1, 1, 2, 3]), chroma_me=rnd.randint(0, 1), mixed_refs=rnd.randint(0, 1), aq_mode=rnd.randint(0, 1), aq_strength=rnd.choice([0.51985, 1.0397, 1.55955]), me_range=rnd.choice([4, 8, 16]),
              qp_i=rnd.randint(0, 51), qp_p=rnd.randint(0, 51), deblock=rnd.randint(0, 1), dct_decimate=rnd.randint(0, 1),
              deblock_alpha=rnd.randint(-3, 3), deblock_beta=rnd.randint(-3, 3), chroma_qp_offset=rnd.randint(-6, 6),
              fast_pskip=rnd.randint(0, 1), mv_range=rnd.choice([0, 0, 32, 64, 128, 512]), cabac=rnd.randint(0, 1))
    if rnd.random() < 0.25 and (h + 15) // 16 >= 8:
        kw["slices"] = rnd.randint(2, (h + 15) // 16 // 4)        # x264 slice threads
    elif rnd.random() < 0.2 and h > 16:
        kw["slices"] = rnd.randint(2, (h + 15) // 16); kw["slices_plain"] = 1          # x264 --slices N: down to one row each, filtered across
    return w, h, kw, rnd.randint(2, 6), rnd.randint(0, 10 ** 6), rnd.random() < 0.3


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_random_configs_bitexact(gpu, seed):
    from gpu_enc import GpuEncoder
    rnd = random.Random(seed)
    for it in range(40):
        w, h, kw, nfr, fseed, second_idr = random_case(rnd)
        frames = synth_frames(w, h, nfr, seed=fseed)
        cfg = O.default_config(w, h, **kw)
        og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
        for i, f in enumerate(frames):
            st = 2 if i == 0 or (i == 3 and second_idr) else 0
            o_mb, o_lv = og.encode(f, st)
            g_mb, g_lv = gg.encode([f], st)
            tag = f"seed {seed} case {it}: {w}x{h} {kw} frame {i}"
            assert np.array_equal(g_mb[0].view(np.uint8), o_mb.view(np.uint8)), tag + " records"
            assert np.array_equal(g_lv[0], o_lv), tag + " levels"
            assert np.array_equal(gg.recon(0), og.recon()), tag + " reconstruction"
        og.close(); gg.close()


@pytest.mark.parametrize("seed", [41, 42, 43])
def test_random_cabac_rd_configs_bitexact(gpu, seed):
    """... and with RD in CABAC sessions (context variables carried on the device, size-only pricing); the context variables the last slice
    ends with are compared too"""
    from gpu_enc import GpuEncoder
    from test_gpu_pipeline import CABAC_CTX_I, CABAC_CTX_P
    rnd = random.Random(seed)
    for it in range(30):
        w, h, kw, nfr, fseed, second_idr = random_case(rnd)
        psy = rnd.randint(0, 1)
        kw.update(cabac=1, rd=1, subme=rnd.choice([6, 7]), psy=psy, psy_rd_q8=rnd.choice([26, 102, 256, 512]) if psy else 0,
                  trellis=rnd.choice([0, 63, 63, 127, rnd.randint(1, 62), 64 + rnd.randint(1, 63)]))          # x264 --trellis 1 is 63 (every site of the final encode), --trellis 2 is 127
        frames = synth_frames(w, h, nfr, seed=fseed)
        cfg = O.default_config(w, h, **kw)
        og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
        for i, f in enumerate(frames):
            st = 2 if i == 0 or (i == 3 and second_idr) else 0
            o_mb, o_lv = og.encode(f, st)
            g_mb, g_lv = gg.encode([f], st)
            tag = f"seed {seed} case {it}: {w}x{h} {kw} frame {i}"
            assert np.array_equal(g_mb[0].view(np.uint8), o_mb.view(np.uint8)), tag + " records"
            assert np.array_equal(g_lv[0], o_lv), tag + " levels"
            assert np.array_equal(gg.recon(0), og.recon()), tag + " reconstruction"
            used = CABAC_CTX_I if st == 2 else CABAC_CTX_P
            assert np.array_equal(gg.cabac_states(0, kw.get("slices", 1) - 1)[used], og.cabac_states()[used]), tag + " context variables"
        og.close(); gg.close()


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_random_rd_configs_bitexact(gpu, seed):
    """the same sweep with RD mode decision on (x264 subme 6 / 7 in a CAVLC session: candidates coded and costed as SSD + psy + lambda2 x bits
    on the device), psy-RD strengths from off to 2.0"""
    from gpu_enc import GpuEncoder
    rnd = random.Random(seed)
    for it in range(30):
        w, h, kw, nfr, fseed, second_idr = random_case(rnd)
        psy = rnd.randint(0, 1)
        kw.update(cabac=0, rd=1, subme=rnd.choice([6, 7]), psy=psy, psy_rd_q8=rnd.choice([26, 102, 256, 512]) if psy else 0)
        frames = synth_frames(w, h, nfr, seed=fseed)
        cfg = O.default_config(w, h, **kw)
        og, gg = O.OracleEncoder(cfg), GpuEncoder(cfg)
        for i, f in enumerate(frames):
            st = 2 if i == 0 or (i == 3 and second_idr) else 0
            o_mb, o_lv = og.encode(f, st)
            g_mb, g_lv = gg.encode([f], st)
            tag = f"seed {seed} case {it}: {w}x{h} {kw} frame {i}"
            assert np.array_equal(g_mb[0].view(np.uint8), o_mb.view(np.uint8)), tag + " records"
            assert np.array_equal(g_lv[0], o_lv), tag + " levels"
            assert np.array_equal(gg.recon(0), og.recon()), tag + " reconstruction"
        og.close(); gg.close()


@pytest.mark.parametrize("seed", [21, 22])
def test_random_multi_stream_quantisers_bitexact(gpu, seed):
    """several streams in lock-step, random toolsets, and per picture a random choice of quantiser source: the shared pair, one slice
    quantiser per stream (x264gpu_encoder_set_stream_qps), per-macroblock offsets from the caller (set_mb_qp_offsets), or both at
    once — every stream must equal its own oracle encoder"""
    import torch
    from gpu_enc import GpuEncoder
    from x264vfw_amd import lib
    rnd = random.Random(seed)
    for it in range(12):
        w, h, kw, nfr, fseed, _ = random_case(rnd)
        S = rnd.randint(2, 4)
        kw["qp_i"], kw["qp_p"] = rnd.randint(10, 44), rnd.randint(10, 44)
        seqs = [synth_frames(w, h, nfr, seed=fseed + 17 * s) for s in range(S)]
        ogs = [O.OracleEncoder(O.default_config(w, h, **kw)) for _ in range(S)]
        gg = GpuEncoder(O.default_config(w, h, streams=S, **kw))
        n = ogs[0].n
        nprng = np.random.default_rng(fseed)
        keep = []
        for i in range(nfr):
            st = 2 if i == 0 else 0
            mode = rnd.choice(["shared", "stream", "offsets", "both"])
            qps = [rnd.randint(8, 46) for _ in range(S)] if mode in ("stream", "both") else None
            off = (nprng.integers(-1500, 1500, (S, n)) / np.float32(256) + nprng.random((S, n), np.float32) / 64).astype(np.float32) if mode in ("offsets", "both") else None
            arr = None if qps is None else np.array(qps, np.int8)
            lib.check(lib.x264gpu_encoder_set_stream_qps(gg.h, None if arr is None else arr.ctypes.data), "set_stream_qps")
            d_off = None if off is None else torch.from_numpy(off.copy()).cuda()
            keep.append(d_off)
            lib.check(lib.x264gpu_encoder_set_mb_qp_offsets(gg.h, None if d_off is None else d_off.data_ptr()), "set_mb_qp_offsets")
            g_mb, g_lv = gg.encode([seqs[s][i] for s in range(S)], st)
            for s in range(S):
                q = qps[s] if qps else (kw["qp_i"] if st == 2 else kw["qp_p"])
                ogs[s].set_qp(q, q)
                ogs[s].set_mb_qp_offsets(None if off is None else off[s])
                o_mb, o_lv = ogs[s].encode(seqs[s][i], st)
                tag = f"seed {seed} case {it}: {w}x{h} S={S} {kw} frame {i} stream {s} mode {mode}"
                assert np.array_equal(g_mb[s].view(np.uint8), o_mb.view(np.uint8)), tag + " records"
                assert np.array_equal(g_lv[s], o_lv), tag + " levels"
                assert np.array_equal(gg.recon(s), ogs[s].recon()), tag + " reconstruction"
        for o in ogs:
            o.close()
        gg.close()


@pytest.mark.parametrize("seed", [3, 57])
def test_random_b_picture_configs_bitexact(gpu, seed):
    """randomised B-picture streams (tools/fuzz_soak_b.py random_b_case: sizes, runs of B pictures with and without b-pyramid, references, search
    methods and ranges, p8x8 / b8x8 apart, weightb, trellis 0 / 1 / 2, psy strengths, slices, --weightp 2): device = CPU checker picture by picture,
    and the device's stream decodes to its reconstruction.  Seed 57's first case is the one a 1 250-case soak of this round found: the B search
    must not move the half-pel threshold by the reference cost (x264's P search does, its B search does not)"""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fuzz_soak_b import random_b_case
    from test_gpu_bframes import run
    rnd = random.Random(seed)
    for it in range(12):
        w, h, types, fseed, bframes, pyramid, weightp, kw = random_b_case(rnd)
        run(gpu, w, h, types, fseed, bframes=bframes, pyramid=pyramid, weightp=weightp, weights=kw.pop("_weights", None), qp_frac=kw.pop("_qp_frac", None), **kw)

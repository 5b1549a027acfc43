"""-m gpu: Tier-1 primitives of libx264gpu.so vs the CPU oracle, bit-exact, through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def blocks(rng, n, h, w, kind):
    if kind == "rand":
        return rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    if kind == "extreme":
        return rng.choice(np.array([0, 255], np.uint8), (n, h, w))
    base = rng.integers(0, 256, (n, 1, 1))
    return np.clip(base + rng.integers(-6, 7, (n, h, w)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("metric,name", [(0, "sad"), (1, "satd"), (2, "sa8d"), (3, "ssd")])
@pytest.mark.parametrize("w,h", [(16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)])
def test_pixel_metric(gpu, metric, name, w, h):
    import torch
    if name == "sa8d" and (w, h) not in ((16, 16), (8, 8)):
        pytest.skip("sa8d is defined for 8x8 and 16x16")
    rng = np.random.default_rng(1000 * metric + 16 * w + h)
    for kind in ("rand", "extreme", "smooth"):
        n = 257
        a, b = blocks(rng, n, h, w, kind), blocks(rng, n, h, w, kind)
        da, db = dev(a), dev(b)
        out = torch.empty(n, dtype=torch.int32, device="cuda")
        gpu.check(gpu.x264gpu_pixel_metric(metric, da.data_ptr(), db.data_ptr(), n, w, h, out.data_ptr(), None))
        np.testing.assert_array_equal(out.cpu().numpy(), O.metric(name, a, b), err_msg=f"{name} {w}x{h} {kind}")


@pytest.mark.parametrize("w,h", [(16, 16), (8, 8), (8, 16)])
def test_pixel_var(gpu, w, h):
    import torch
    rng = np.random.default_rng(7)
    n = 130
    a = blocks(rng, n, h, w, "rand")
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    da = dev(a)
    gpu.check(gpu.x264gpu_pixel_var(da.data_ptr(), n, w, h, out.data_ptr(), None))
    ref = np.array([O.L.x264o_var(O.ptr(a, i * w * h), w, w, h) for i in range(n)], np.uint64)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), ref)


@pytest.mark.parametrize("w,h", [(16, 16), (8, 8), (8, 16), (16, 8)])
@pytest.mark.parametrize("kind", ["rand", "flat", "extreme"])
def test_pixel_hadamard_ac(gpu, w, h, kind):
    """pixel_hadamard_ac (psy-RD energy): AC sums of the 4x4 and 8x8 Hadamard transforms of source blocks, packed as x264 does"""
    import torch
    rng = np.random.default_rng(11)
    n = 77
    a = blocks(rng, n, h, w, kind)
    out = torch.empty(n, dtype=torch.int64, device="cuda")
    da = dev(a)
    gpu.check(gpu.x264gpu_pixel_hadamard_ac(da.data_ptr(), n, w, h, out.data_ptr(), None))
    ref = np.array([O.L.x264o_hadamard_ac(O.ptr(a, i * w * h), w, w, h) for i in range(n)], np.uint64)
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint64), ref)


@pytest.mark.parametrize("qp", [0, 5, 12, 23, 24, 26, 35, 36, 37, 51])
@pytest.mark.parametrize("lst", [0, 1, 2, 3])
def test_dctq4x4(gpu, qp, lst):
    import torch
    rng = np.random.default_rng(qp * 4 + lst)
    n = 1000
    enc = np.concatenate([blocks(rng, n // 2, 4, 4, "rand"), blocks(rng, n // 2, 4, 4, "smooth")])
    pred = np.concatenate([blocks(rng, n // 2, 4, 4, "smooth"), blocks(rng, n // 2, 4, 4, "extreme")])
    coef = torch.empty((n, 16), dtype=torch.int16, device="cuda")
    lev = torch.empty_like(coef)
    rec = torch.empty((n, 16), dtype=torch.uint8, device="cuda")
    denc, dpred = dev(enc), dev(pred)   # keep alive: freed temporaries would alias in the caching allocator
    gpu.check(gpu.x264gpu_dctq4x4(denc.data_ptr(), dpred.data_ptr(), n, qp, lst, coef.data_ptr(),
                                  lev.data_ptr(), rec.data_ptr(), None))
    rc, rl, rr = O.dctq4x4(enc, pred, qp, lst)
    np.testing.assert_array_equal(coef.cpu().numpy(), rc)
    np.testing.assert_array_equal(lev.cpu().numpy(), rl)
    np.testing.assert_array_equal(rec.cpu().numpy().reshape(n, 4, 4), rr)


@pytest.mark.parametrize("w,h", [(64, 48), (100, 36), (352, 288)])
def test_hpel_filter(gpu, w, h):
    rng = np.random.default_rng(w + h)
    pad = 32
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    planes, stride = O.make_padded_planes(img, pad)
    ref = planes.copy()
    O.frame_filter(ref, stride, w, h, pad)
    d = dev(planes)
    pb = planes.shape[1] * stride
    gpu.check(gpu.x264gpu_hpel_filter(d.data_ptr(), pb, stride, w, h, pad, None))
    got = d.cpu().numpy()
    for k in range(4):
        np.testing.assert_array_equal(got[k, :, :w + 2 * pad], ref[k, :, :w + 2 * pad], err_msg=f"plane {k}")


@pytest.mark.parametrize("w,h", [(64, 48), (98, 34), (352, 288)])
def test_lowres(gpu, w, h):
    import torch
    rng = np.random.default_rng(w * h)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    ds = (w // 2 + 31) // 32 * 32
    ref = np.zeros((4, h // 2, ds), np.uint8)
    arr = (C.c_void_p * 4)(*[O.ptr(ref, k * (h // 2) * ds) for k in range(4)])
    O.L.x264o_frame_init_lowres(O.ptr(img), w, w, h, arr, ds)
    out = torch.zeros((4, h // 2, ds), dtype=torch.uint8, device="cuda")
    dimg = dev(img)
    gpu.check(gpu.x264gpu_lowres(dimg.data_ptr(), w, w, h, out.data_ptr(), (h // 2) * ds, ds, None))
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_mc_luma_chroma(gpu):
    import torch
    rng = np.random.default_rng(99)
    w, h, pad = 96, 64, 32
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    planes, stride = O.make_padded_planes(img, pad)
    O.frame_filter(planes, stride, w, h, pad)
    d = dev(planes)
    pb = planes.shape[1] * stride
    p00 = d.data_ptr() + pad * stride + pad
    pp = O.plane_ptrs(planes, stride, pad)
    for bw, bh in ((16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)):
        n = 300
        xy = np.stack([rng.integers(0, w - bw + 1, n), rng.integers(0, h - bh + 1, n)], 1).astype(np.int32)
        mv = rng.integers(-4 * 20, 4 * 20, (n, 2)).astype(np.int32)
        mv[:16] = [[i & 3, i >> 2] for i in range(16)]
        out = torch.empty((n, bh, bw), dtype=torch.uint8, device="cuda")
        dxy, dmv = dev(xy), dev(mv)
        gpu.check(gpu.x264gpu_mc_luma(p00, pb, stride, dxy.data_ptr(), dmv.data_ptr(), n, bw, bh,
                                      out.data_ptr(), None))
        ref = np.empty((n, bh, bw), np.uint8)
        for i in range(n):
            O.L.x264o_mc_luma(O.ptr(ref, i * bw * bh), bw, pp, stride, int(xy[i, 0]), int(xy[i, 1]),
                              int(mv[i, 0]), int(mv[i, 1]), bw, bh)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"mc_luma {bw}x{bh}")
    # chroma: padded NV12 plane, 16 px of padding (32 bytes each side)
    cw, ch, cpad = w // 2, h // 2, 16
    cs = (2 * cw + 4 * cpad + 63) // 64 * 64
    nv = rng.integers(0, 256, (ch + 2 * cpad, cs), dtype=np.uint8)
    dnv = dev(nv)
    org = cpad * cs + 2 * cpad
    for bw, bh in ((8, 8), (8, 4), (4, 8), (4, 4)):
        n = 300
        xy = np.stack([rng.integers(0, cw - bw + 1, n), rng.integers(0, ch - bh + 1, n)], 1).astype(np.int32)
        mv = rng.integers(-8 * 10, 8 * 10, (n, 2)).astype(np.int32)
        out = torch.empty((n, 2, bh, bw), dtype=torch.uint8, device="cuda")
        dxy, dmv = dev(xy), dev(mv)
        gpu.check(gpu.x264gpu_mc_chroma(dnv.data_ptr() + org, cs, dxy.data_ptr(), dmv.data_ptr(), n, bw, bh,
                                        out.data_ptr(), None))
        ref = np.empty((n, 2, bh, bw), np.uint8)
        for i in range(n):
            O.L.x264o_mc_chroma(O.ptr(ref, i * 2 * bw * bh), O.ptr(ref, i * 2 * bw * bh + bw * bh), bw,
                                O.ptr(nv, org), cs, int(xy[i, 0]), int(xy[i, 1]), int(mv[i, 0]), int(mv[i, 1]), bw, bh)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"mc_chroma {bw}x{bh}")


@pytest.mark.parametrize("qp", [0, 7, 23, 35, 36, 42, 51])
@pytest.mark.parametrize("lst", [0, 1])
def test_dctq8x8(gpu, qp, lst):
    import torch
    rng = np.random.default_rng(800 + qp * 2 + lst)
    n = 333
    enc = np.concatenate([blocks(rng, n - 100, 8, 8, "rand"), blocks(rng, 100, 8, 8, "smooth")])
    pred = np.concatenate([blocks(rng, n - 100, 8, 8, "smooth"), blocks(rng, 100, 8, 8, "extreme")])
    coef = torch.empty((n, 64), dtype=torch.int16, device="cuda")
    lev = torch.empty_like(coef)
    rec = torch.empty((n, 64), dtype=torch.uint8, device="cuda")
    denc, dpred = dev(enc), dev(pred)
    gpu.check(gpu.x264gpu_dctq8x8(denc.data_ptr(), dpred.data_ptr(), n, qp, lst, coef.data_ptr(), lev.data_ptr(), rec.data_ptr(), None))
    rc, rl, rr = O.dctq8x8(enc, pred, qp, lst)
    np.testing.assert_array_equal(coef.cpu().numpy(), rc)
    np.testing.assert_array_equal(lev.cpu().numpy(), rl)
    np.testing.assert_array_equal(rec.cpu().numpy().reshape(n, 8, 8), rr)


def test_intra_predictors(gpu):
    import torch
    rng = np.random.default_rng(4242)
    W = H = 96
    plane = rng.integers(0, 256, (H, W), dtype=np.uint8)
    dplane = dev(plane)
    for kind, size, nmodes, fn in ((0, 16, 7, O.L.x264o_predict_16x16), (1, 8, 7, O.L.x264o_predict_8x8c)):
        n = 200
        xy = rng.integers(8, W - size - 8, (n, 2)).astype(np.int32)
        mode = rng.integers(0, nmodes, n).astype(np.int32)
        avail = np.zeros(n, np.int32)
        out = torch.empty((n, size, size), dtype=torch.uint8, device="cuda")
        dxy, dmode, dav = dev(xy), dev(mode), dev(avail)
        gpu.check(gpu.x264gpu_intra_predict(kind, dplane.data_ptr(), W, dxy.data_ptr(), dmode.data_ptr(), dav.data_ptr(), n, out.data_ptr(), None))
        ref = np.zeros((n, size, size), np.uint8)
        for i in range(n):
            fn(O.ptr(ref, i * size * size), size, O.ptr(plane, int(xy[i, 1]) * W + int(xy[i, 0])), W, int(mode[i]))
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"kind {kind}")
    # 4x4: the nine modes with every legal availability combination
    n = 600
    xy = rng.integers(8, W - 16, (n, 2)).astype(np.int32)
    mode = rng.integers(0, 9, n).astype(np.int32)
    avail = rng.integers(0, 16, n).astype(np.int32)
    for i in range(n):
        m, a = int(mode[i]), int(avail[i])
        need = {0: 2, 1: 1, 2: 0, 3: 2, 4: 11, 5: 11, 6: 11, 7: 2, 8: 1}[m]
        avail[i] = a | need
        if not avail[i] & 2:
            avail[i] &= ~4
    out = torch.empty((n, 4, 4), dtype=torch.uint8, device="cuda")
    dxy, dmode, dav = dev(xy), dev(mode), dev(avail)
    gpu.check(gpu.x264gpu_intra_predict(2, dplane.data_ptr(), W, dxy.data_ptr(), dmode.data_ptr(), dav.data_ptr(), n, out.data_ptr(), None))
    ref = np.zeros((n, 4, 4), np.uint8)
    for i in range(n):
        m, a = int(mode[i]), int(avail[i])
        real = m
        if m == 2:
            real = 2 if (a & 1 and a & 2) else 9 if a & 1 else 10 if a & 2 else 11
        O.L.x264o_predict_4x4(O.ptr(ref, i * 16), 4, O.ptr(plane, int(xy[i, 1]) * W + int(xy[i, 0])), W, real, a)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)


def test_mc_bipred_avg_and_weight(gpu):
    """A10's sample combiners behind bi-prediction and --weightp: pixel_avg / pixel_avg_weight (implicit weights span -64..128) and
    mc_weight over every denominator, on random sample blocks, vs oracle/mc.c"""
    import torch
    rng = np.random.default_rng(1234)
    n = 16 * 16 * 40
    a = rng.integers(0, 256, n, dtype=np.uint8)
    b = rng.integers(0, 256, n, dtype=np.uint8)
    a[:8] = [0, 255, 0, 255, 1, 254, 128, 127]
    b[:8] = [0, 255, 255, 0, 254, 1, 127, 128]
    da, db = dev(a), dev(b)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    ref = np.empty(n, np.uint8)
    for w1 in (32, 0, 64, 21, 43, -64, 128, -10, 90, 1, 63):
        gpu.check(gpu.x264gpu_mc_avg(da.data_ptr(), db.data_ptr(), n, w1, out.data_ptr(), None))
        O.L.x264o_pixel_avg_weight(O.ptr(ref), 16, O.ptr(a), 16, O.ptr(b), 16, 16, n // 16, w1)
        np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"avg weight {w1}")
    for denom in range(8):
        for scale, offset in ((1 << denom, 0), (1 << denom, -7), (0, 5), (255, -128), (37, 127), ((1 << denom) + 3, 9), (max(1, (1 << denom) - 1), -3)):
            gpu.check(gpu.x264gpu_mc_weight(da.data_ptr(), n, scale, denom, offset, out.data_ptr(), None))
            O.L.x264o_mc_weight(O.ptr(ref), 16, O.ptr(a), 16, 16, n // 16, scale, denom, offset)
            np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"weight scale {scale} denom {denom} offset {offset}")
    assert gpu.x264gpu_mc_avg(da.data_ptr(), db.data_ptr(), 6, 32, out.data_ptr(), None) < 0          # not a multiple of four samples
    assert gpu.x264gpu_mc_weight(da.data_ptr(), n, 1, 8, 0, out.data_ptr(), None) < 0                  # log2 denominator is 0..7


@pytest.mark.parametrize("alpha,beta,cqo", [(0, 0, 0), (2, -1, -3), (-3, 3, 5)])
def test_deblock_primitive_vs_the_standard(gpu, alpha, beta, cqo):
    """A9 on its own: the in-loop filter kernel of the frame pipeline on given pictures + random macroblock records (intra / inter / skip
    mixes, per-macroblock quantisers, coded-block bits, 8x8 transform flags, two references, vectors around the 4-quarter-sample
    threshold) against a per-sample restatement of 8.7 (tests/spec_ref.py deblock_picture: bS derivation, edge order, qp averaging,
    chroma quantiser table — no oracle code involved)"""
    import torch
    import spec_ref as S
    w, h, streams = 96, 64, 2
    mbw, mbh = w // 16, h // 16
    rng = np.random.default_rng(100 + alpha * 7 + beta)
    cfg = O.default_config(w, h, streams=streams, deblock=1, deblock_alpha=alpha, deblock_beta=beta, chroma_qp_offset=cqo)
    h_ = C.c_void_p()
    gpu.check(gpu.x264gpu_encoder_create(C.byref(h_), C.byref(cfg)), "create")
    pics, recs, want = [], [], []
    for s in range(streams):
        # blocky pictures: each 4x4 block has its own level near its neighbours' + a little noise, so that every filter branch is reached
        lvl = np.cumsum(rng.integers(-6, 7, (h // 4, w // 4)), axis=1) + np.cumsum(rng.integers(-5, 6, (h // 4, 1)), axis=0) + 120
        Y = np.clip(np.kron(lvl, np.ones((4, 4), np.int64)) + rng.integers(-2, 3, (h, w)), 0, 255)
        U = np.clip(np.kron(lvl[::2, ::2] // 2 + 64, np.ones((4, 4), np.int64)) + rng.integers(-1, 2, (h // 2, w // 2)), 0, 255)
        V = np.clip(255 - U + rng.integers(-3, 4, U.shape), 0, 255)
        mbs = np.zeros(mbw * mbh, O.MB_DTYPE)
        for m in mbs:
            m["type"] = rng.choice([0, 1, 2, 4, 4, 4, 5, 5, 6, 6])
            m["qp"] = rng.integers(8, 50)
            m["transform8x8"] = 1 if m["type"] == 1 else (rng.integers(0, 2) if m["type"] in (4, 5) else 0)
            m["nnz"] = rng.integers(0, 1 << 16) & rng.integers(0, 1 << 16)
            m["cbp_luma"] = sum(1 << i for i in range(4) if (int(m["nnz"]) >> (4 * i)) & 15) if m["type"] != 6 else 0
            if m["type"] >= 4:
                m["ref"] = rng.integers(0, 2, 4) if m["type"] == 5 else [rng.integers(0, 2)] * 4
                base = rng.integers(-9, 10, 2)
                m["mv"] = base + (rng.integers(-4, 5, (4, 2)) if m["type"] == 5 else 0)
            else:
                m["ref"] = -1
            if m["type"] == 6:
                m["nnz"] = 0
                m["ref"] = 0
        pics.append(np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)]).astype(np.uint8))
        recs.append(mbs)
        S.deblock_picture(Y, U, V, mbs, mbw, mbh, alpha, beta, cqo)
        want.append(np.concatenate([Y.reshape(-1), U.reshape(-1), V.reshape(-1)]).astype(np.uint8))
    d_in = torch.from_numpy(np.stack(pics)).cuda()
    d_mb = torch.from_numpy(np.stack(recs).view(np.uint8).reshape(streams, -1)).cuda()
    d_out = torch.empty_like(d_in)
    gpu.check(gpu.x264gpu_encoder_deblock_pictures(h_, d_in.data_ptr(), d_mb.data_ptr(), d_out.data_ptr(), None), "deblock_pictures")
    got = d_out.cpu().numpy()
    gpu.x264gpu_encoder_destroy(h_)
    for s in range(streams):
        assert not np.array_equal(want[s], pics[s])                  # the filter did something
        np.testing.assert_array_equal(got[s], want[s], err_msg=f"stream {s}")


@pytest.mark.parametrize("cat", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("qp,intra", [(23, 0), (20, 1), (8, 0), (37, 1), (51, 0), (30, 0)])
def test_trellis_primitive_vs_oracle(gpu, cat, qp, intra):
    """x264's CABAC trellis quantiser as a device primitive (csrc/trellis.hip.h: eight lanes per block = the eight nodes of the search) against
    the checker's restatement (oracle/trellis.cpp), block by block: random transform coefficients with a natural spectrum, random context
    variables, every block category, quantisers from 8 to 51, inter and intra lambda.  The levels must be identical"""
    import ctypes as C
    import torch
    from x264vfw_amd import lib
    rng = np.random.default_rng(1000 * cat + qp + intra)
    nc = 64 if cat == 5 else 4 if cat == 3 else 16
    nblk = 203
    zz4 = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]
    zz8 = np.zeros(64, np.int64)
    r, c_, up = 0, 0, True
    for i in range(64):                       # frame zigzag of an 8x8 block
        zz8[i] = r * 8 + c_
        if up:
            if c_ == 7: r += 1; up = False
            elif r == 0: c_ += 1; up = False
            else: r -= 1; c_ += 1
        else:
            if r == 7: c_ += 1; up = True
            elif c_ == 0: r += 1; up = True
            else: r += 1; c_ -= 1
    zz = zz8 if cat == 5 else np.arange(4) if cat == 3 else np.array(zz4)
    # quantiser rows as oracle/quant.c builds them (flat matrices)
    s4 = [[13107, 8066, 5243], [11916, 7490, 4660], [10082, 6554, 4194], [9362, 5825, 3647], [8192, 5243, 3355], [7282, 4559, 2893]]
    s8 = [[13107, 11428, 20972, 12222, 16777, 15481], [11916, 10826, 19174, 11058, 14980, 14290], [10082, 8943, 15978, 9675, 12710, 11985],
          [9362, 8228, 14913, 8931, 11984, 11259], [8192, 7346, 13159, 7740, 10486, 9777], [7282, 6428, 11570, 6830, 9118, 8640]]
    shr = lambda x, s: x << -s if s <= 0 else (x + (1 << (s - 1))) >> s
    cls8 = [[0, 3, 4, 3], [3, 1, 5, 1], [4, 5, 2, 5], [3, 1, 5, 1]]
    if cat == 5:
        mf = np.array([shr(s8[qp % 6][cls8[(i >> 3) & 3][i & 3]], qp // 6) for i in range(64)], np.uint16)
    else:
        mf = np.array([shr(s4[qp % 6][(i & 1) + ((i >> 2) & 1)], qp // 6 - 1) for i in range(16)], np.uint16)
    step = 65536.0 / float(mf[0])
    # coefficients: Laplacian amplitudes falling with frequency, a share of blocks nearly empty, a few with big levels
    scan_pos = np.arange(nc)
    amp = step * (2.5 / (1.0 + 0.35 * scan_pos))
    coefs_scan = (rng.laplace(0.0, 1.0, (nblk, nc)) * amp * rng.choice([0.15, 0.6, 1.0, 3.0, 12.0], (nblk, 1))).astype(np.int64)
    coefs_scan = np.clip(coefs_scan, -30000, 30000).astype(np.int16)
    if cat in (1, 4):
        coefs_scan[:, 0] = 0
    coefs_scan[0] = 0                                   # an empty block
    coefs_scan[1] = 0; coefs_scan[1, 1 if cat in (1, 4) else 0] = int(step * 1.4)   # only the first coefficient
    states = ((rng.integers(0, 63, 460) << 1) | rng.integers(0, 2, 460)).astype(np.uint8)
    O.L.x264o_quant_trellis_cabac.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    O.L.x264o_quant_trellis_cabac.restype = C.c_int
    want = np.zeros_like(coefs_scan)
    want_nz = np.zeros(nblk, np.uint8)
    for b in range(nblk):
        raster = np.zeros(nc, np.int16)
        raster[zz] = coefs_scan[b]
        want_nz[b] = O.L.x264o_quant_trellis_cabac(raster.ctypes.data, mf.ctypes.data, qp, cat, intra, states.ctypes.data) != 0
        want[b] = raster[zz]
    d_c, d_s = torch.from_numpy(coefs_scan.copy()).cuda(), torch.from_numpy(states).cuda()
    d_l, d_z = torch.zeros_like(d_c), torch.zeros(nblk, dtype=torch.uint8, device="cuda")
    lib.check(lib.x264gpu_trellis_blocks(d_c.data_ptr(), nblk, cat, qp, intra, d_s.data_ptr(), d_l.data_ptr(), d_z.data_ptr(), None), "trellis_blocks")
    torch.cuda.synchronize()
    got = d_l.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert len(bad) == 0, f"{len(bad)} of {nblk} blocks differ; first {bad[0]}: coefs {coefs_scan[bad[0]].tolist()} device {got[bad[0]].tolist()} oracle {want[bad[0]].tolist()}"
    np.testing.assert_array_equal(d_z.cpu().numpy(), want_nz)
    assert np.count_nonzero(want) > nblk          # the cases are not trivial


def test_cabac_level_walk_primitive_vs_serial_restatement(gpu):
    """the level walk of the CABAC pricing (csrc/cabac_rd.hip.h cab_levels_all: every block of a macroblock at once, one chain per context) against the
    block-by-block, bin-by-bin restatement of x264's coder (tests/cabac_levels_ref.py): same bits, same context variables — every luma category,
    chroma DC / AC, sparse and dense blocks, levels into the escape range, partial block masks"""
    import torch
    import cabac_levels_ref as R
    lib = gpu
    cases = R.random_cases(1500, 0xcab)
    n = len(cases)
    lv = np.array([c[0] for c in cases], np.int16)
    what = np.array([[c[1]["cat0"], c[1]["nz0"], c[1]["nzac"], c[1]["nzdc"], c[1]["ldc"]] for c in cases], np.int32)
    pack = lambda rows: np.array([[b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24) for b in r] for r in rows], np.uint32)
    r_in, r8_in = pack([c[2] for c in cases]), np.array([c[3] for c in cases], np.uint32)
    r_want, r8_want = pack([c[4] for c in cases]), np.array([c[5] for c in cases], np.uint32)
    bits_want = np.array([c[6] for c in cases], np.int32)
    d = lambda a: torch.from_numpy(a.copy()).cuda()
    d_lv, d_w, d_r, d_r8 = d(lv), d(what), d(r_in.view(np.int32)), d(r8_in.view(np.int32))
    d_ro, d_r8o, d_b = torch.zeros_like(d_r), torch.zeros_like(d_r8), torch.zeros(n, dtype=torch.int32, device="cuda")
    lib.check(lib.x264gpu_cabac_level_walk(d_lv.data_ptr(), d_w.data_ptr(), n, d_r.data_ptr(), d_r8.data_ptr(), d_ro.data_ptr(), d_r8o.data_ptr(), d_b.data_ptr(), None), "cabac_level_walk")
    torch.cuda.synchronize()
    got_b, got_r, got_r8 = d_b.cpu().numpy(), d_ro.cpu().numpy().view(np.uint32), d_r8o.cpu().numpy().view(np.uint32)
    bad = np.nonzero(got_b != bits_want)[0]
    assert len(bad) == 0, f"bits differ in {len(bad)} of {n} cases; first {bad[0]}: what {what[bad[0]].tolist()} device {got_b[bad[0]]} serial {bits_want[bad[0]]}"
    badr = np.nonzero((got_r != r_want).any(axis=1) | (got_r8 != r8_want).any(axis=1))[0]
    assert len(badr) == 0, f"context variables differ in {len(badr)} of {n} cases; first {badr[0]}: what {what[badr[0]].tolist()}"
    assert bits_want.max() > 100000 and (what[:, 0] == 5).sum() > 100


@pytest.mark.parametrize("streams,nmb,density", [(3, 396, 0.05), (2, 8160, 0.02), (5, 17, 0.0), (1, 99, 1.0), (4, 120, 0.5)])
def test_pack_levels_in_place_vs_numpy(gpu, streams, nmb, density):
    """x264gpu_pack_levels (one wavefront a stream, in place): the groups of 16 levels that hold a non-zero one, one behind the other in macroblock and group order,
    and the index that finds them — against a numpy restatement; empty pictures, dense ones (nothing moves), 1080p's macroblock count; host/host.hpp mb_levels is the
    inverse (tests/test_host_cpu.py packs through the stub and decodes the streams)"""
    import torch
    lib = gpu
    L = 416
    rng = np.random.default_rng(7 * streams + nmb)
    lv = np.zeros((streams, nmb, L // 16, 16), np.int16)
    on = rng.random((streams, nmb, L // 16)) < density
    vals = rng.integers(-300, 300, lv.shape).astype(np.int16)
    vals[vals == 0] = 1
    sparse = rng.random(lv.shape) < 0.3          # a kept group holds a few non-zero levels, not sixteen
    lv[on] = (vals * sparse)[on]
    lv[on, 5] |= 1                                # ... at least one
    want_ix = np.zeros((streams, nmb, 2), np.uint32)
    want_lv = []
    for s in range(streams):
        at, out = 0, []
        for i in range(nmb):
            g = np.nonzero(lv[s, i].any(axis=1))[0]
            want_ix[s, i] = (at, sum(1 << int(k) for k in g))
            out += [lv[s, i, k] for k in g]
            at += len(g)
        want_lv.append(np.array(out, np.int16).reshape(-1))
    d_lv = torch.from_numpy(lv.reshape(-1).copy()).cuda()
    d_ix = torch.zeros(streams * nmb * 2, dtype=torch.int32, device="cuda")
    d_kept = torch.zeros(streams, dtype=torch.int32, device="cuda")
    lib.check(lib.x264gpu_pack_levels(d_lv.data_ptr(), streams, nmb, d_ix.data_ptr(), d_kept.data_ptr(), None), "pack_levels")
    torch.cuda.synchronize()
    assert d_kept.cpu().numpy().tolist() == [len(x) // 16 for x in want_lv]
    got_ix = d_ix.cpu().numpy().view(np.uint32).reshape(streams, nmb, 2)
    got_lv = d_lv.cpu().numpy().reshape(streams, -1)
    assert np.array_equal(got_ix, want_ix)
    for s in range(streams):
        assert np.array_equal(got_lv[s, :len(want_lv[s])], want_lv[s]), s
    assert density == 0.0 or sum(len(x) for x in want_lv) > 0

"""CPU: pin the oracle's NORMATIVE operations against the independent numpy restatement of ITU-T H.264
(tests/spec_ref.py).  The reference ships no tests or golden vectors, so the standard's
own definitions are the known-answer source for everything a decoder must reproduce."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import spec_ref as S

L = O.L


def rnd_res(rng, n, amp=255):
    return rng.integers(-amp, amp + 1, (n, n)).astype(np.int64)


def test_fwd4_matches_matrix_form():
    rng = np.random.default_rng(1)
    for _ in range(200):
        enc = rng.integers(0, 256, (4, 4), dtype=np.uint8)
        pred = rng.integers(0, 256, (4, 4), dtype=np.uint8)
        d = np.zeros(16, np.int16)
        L.x264o_sub4x4_dct(d, O.ptr(enc), 4, O.ptr(pred), 4)
        np.testing.assert_array_equal(d.reshape(4, 4), S.fwd4(enc.astype(np.int64) - pred))


def test_inv4_matches_spec():
    rng = np.random.default_rng(2)
    for _ in range(300):
        c = rng.integers(-2000, 2001, (4, 4)).astype(np.int16)
        base = rng.integers(0, 256, (4, 4), dtype=np.uint8)
        dst = base.copy()
        L.x264o_add4x4_idct(O.ptr(dst), 4, c.reshape(-1).copy())
        np.testing.assert_array_equal(dst, np.clip(base + S.inv4(c), 0, 255))


def test_inv8_matches_spec():
    rng = np.random.default_rng(3)
    for _ in range(100):
        c = rng.integers(-1500, 1501, (8, 8)).astype(np.int16)
        base = rng.integers(0, 256, (8, 8), dtype=np.uint8)
        dst = base.copy()
        L.x264o_add8x8_idct8(O.ptr(dst), 8, c.reshape(-1).copy())
        np.testing.assert_array_equal(dst, np.clip(base + S.inv8(c), 0, 255))


@pytest.mark.parametrize("qp", [0, 11, 23, 24, 30, 35, 36, 44, 51])
def test_dequant_matches_spec(qp):
    rng = np.random.default_rng(qp)
    t = O.quant_tables()
    for _ in range(30):
        c4 = rng.integers(-60, 61, (4, 4)).astype(np.int16)
        d = c4.reshape(-1).copy()
        L.x264o_dequant_4x4(d, C.addressof(t.dequant4_mf), qp)
        np.testing.assert_array_equal(d.reshape(4, 4), S.dequant4(c4, qp).astype(np.int16))
        c8 = rng.integers(-40, 41, (8, 8)).astype(np.int16)
        d = c8.reshape(-1).copy()
        L.x264o_dequant_8x8(d, C.addressof(t.dequant8_mf), qp)
        np.testing.assert_array_equal(d.reshape(8, 8), S.dequant8(c8, qp).astype(np.int16))
        dc = rng.integers(-30, 31, (4, 4)).astype(np.int16)
        d = dc.reshape(-1).copy()
        L.x264o_idct4x4dc(d)
        L.x264o_dequant_4x4_dc(d, C.addressof(t.dequant4_mf), qp)
        np.testing.assert_array_equal(d.reshape(4, 4), S.luma_dc_dequant(dc, qp).astype(np.int16))
        cd = rng.integers(-30, 31, (2, 2)).astype(np.int16)
        out = np.zeros(4, np.int16)
        L.x264o_dequant_2x2_dc(out, cd.reshape(-1).copy(), C.addressof(t.dequant4_mf), qp)
        np.testing.assert_array_equal(out.reshape(2, 2), S.chroma_dc_dequant(cd, qp).astype(np.int16))


@pytest.mark.parametrize("qp", [0, 10, 20, 26, 34, 40])
def test_transform_quant_roundtrip_is_bounded(qp):
    """non-normative forward transform + quant, checked through the normative inverse: reconstruction
    error stays within the quantiser step for every 4x4 and 8x8 basis position (catches table typos)"""
    rng = np.random.default_rng(100 + qp)
    t = O.quant_tables()
    step = 0.625 * 2 ** (qp / 6)
    for _ in range(60):
        pred = np.full((8, 8), 128, np.uint8)
        enc = np.clip(128 + rng.integers(-100, 101, (8, 8)), 0, 255).astype(np.uint8)
        _, _, rec4 = O.dctq4x4(enc[:4, :4].copy().reshape(1, 4, 4), pred[:4, :4].copy().reshape(1, 4, 4), qp, 0)
        assert np.abs(rec4[0].astype(int) - enc[:4, :4]).max() <= 1.6 * step + 2
        d = np.zeros(64, np.int16)
        L.x264o_sub8x8_dct8(d, O.ptr(enc), 8, O.ptr(pred), 8)
        L.x264o_quant_8x8(d, C.addressof(t.quant8_mf[0][qp]), C.addressof(t.quant8_bias[0][qp]))
        L.x264o_dequant_8x8(d, C.addressof(t.dequant8_mf), qp)
        rec = pred.copy()
        L.x264o_add8x8_idct8(O.ptr(rec), 8, d)
        assert np.abs(rec.astype(int) - enc).max() <= 1.6 * step + 2


def test_dc_transforms():
    rng = np.random.default_rng(5)
    for _ in range(50):
        m = rng.integers(-4000, 4001, (4, 4)).astype(np.int16)
        d = m.reshape(-1).copy()
        L.x264o_dct4x4dc(d)
        np.testing.assert_array_equal(d.reshape(4, 4), ((S.H4 @ m.astype(np.int64) @ S.H4) + 1) >> 1)
        c = rng.integers(-3000, 3001, (2, 2)).astype(np.int16)
        d = c.reshape(-1).copy()
        L.x264o_dct2x2dc(d)
        np.testing.assert_array_equal(d.reshape(2, 2), S.H2 @ c.astype(np.int64) @ S.H2)


def _canvas(rng, n=40):
    return rng.integers(0, 256, (n, n), dtype=np.uint8)


def test_pred4x4_matches_spec():
    rng = np.random.default_rng(6)
    for _ in range(40):
        img = _canvas(rng)
        for mode in range(12):
            for tr in (0, 1):
                out = np.zeros((4, 4), np.uint8)
                avail = 1 | 2 | 8 | (4 if tr else 0)
                L.x264o_predict_4x4(O.ptr(out), 4, O.ptr(img, 8 * 40 + 8), 40, mode, avail)
                np.testing.assert_array_equal(out, S.pred4x4(img, 8, 8, mode, tr), err_msg=f"mode {mode} tr {tr}")


def test_pred16x16_and_chroma_match_spec():
    rng = np.random.default_rng(7)
    for _ in range(25):
        img = _canvas(rng)
        for mode in range(7):
            out = np.zeros((16, 16), np.uint8)
            L.x264o_predict_16x16(O.ptr(out), 16, O.ptr(img, 8 * 40 + 8), 40, mode)
            np.testing.assert_array_equal(out, S.pred16x16(img, 8, 8, mode), err_msg=f"16x16 mode {mode}")
            outc = np.zeros((8, 8), np.uint8)
            L.x264o_predict_8x8c(O.ptr(outc), 8, O.ptr(img, 8 * 40 + 8), 40, mode)
            np.testing.assert_array_equal(outc, S.pred_chroma8x8(img, 8, 8, mode), err_msg=f"chroma mode {mode}")


def test_pred8x8_matches_spec():
    rng = np.random.default_rng(8)
    for _ in range(25):
        img = _canvas(rng)
        for tl in (0, 1):
            for tr in (0, 1):
                edge = np.zeros(33, np.uint8)
                L.x264o_predict_8x8_filter(O.ptr(img, 8 * 40 + 8), 40, O.ptr(edge), 1 | 2 | (8 if tl else 0) | (4 if tr else 0))
                for mode in range(9):
                    if mode in (4, 5, 6) and not tl:
                        continue
                    out = np.zeros((8, 8), np.uint8)
                    L.x264o_predict_8x8(O.ptr(out), 8, O.ptr(edge), mode)
                    np.testing.assert_array_equal(out, S.pred8x8(img, 8, 8, mode, tl, tr), err_msg=f"8x8 mode {mode} tl {tl} tr {tr}")


def test_luma_and_chroma_interpolation_match_spec():
    """hpel planes + qpel averaging (mc_luma) == direct evaluation of 8.4.2.2.1 at all 16 positions,
    including positions outside the picture (clamping == replicated border)"""
    rng = np.random.default_rng(9)
    w, h, pad = 24, 20, 32
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    planes, stride = O.make_padded_planes(img, pad)
    O.frame_filter(planes, stride, w, h, pad)
    pp = O.plane_ptrs(planes, stride, pad)
    out = np.zeros((4, 4), np.uint8)
    for (bx, by) in ((0, 0), (10, 8), (20, 16), (-6, -7), (22, 18)):
        for mvy in range(-5, 9):
            for mvx in range(-5, 9):
                L.x264o_mc_luma(O.ptr(out), 4, pp, stride, bx, by, mvx, mvy, 4, 4)
                exp = np.array([[S.luma_sample(img, bx + x + (mvx >> 2), by + y + (mvy >> 2), mvx & 3, mvy & 3) for x in range(4)] for y in range(4)])
                np.testing.assert_array_equal(out, exp, err_msg=f"blk {bx},{by} mv {mvx},{mvy}")
    cw, ch = 16, 12
    nv = rng.integers(0, 256, (ch + 2, 2 * (cw + 2)), dtype=np.uint8)
    u, v = nv[:, 0::2], nv[:, 1::2]
    ou, ov = np.zeros((4, 4), np.uint8), np.zeros((4, 4), np.uint8)
    for mvy in range(0, 17, 3):
        for mvx in range(0, 17, 3):
            L.x264o_mc_chroma(O.ptr(ou), O.ptr(ov), 4, O.ptr(nv), nv.shape[1], 2, 1, mvx, mvy, 4, 4)
            eu = np.array([[S.chroma_sample(u, 2 + x + (mvx >> 3), 1 + y + (mvy >> 3), mvx & 7, mvy & 7) for x in range(4)] for y in range(4)])
            ev = np.array([[S.chroma_sample(v, 2 + x + (mvx >> 3), 1 + y + (mvy >> 3), mvx & 7, mvy & 7) for x in range(4)] for y in range(4)])
            np.testing.assert_array_equal(ou, eu)
            np.testing.assert_array_equal(ov, ev)


def test_deblock_lines_match_spec():
    rng = np.random.default_rng(10)
    for it in range(4000):
        base = int(rng.integers(20, 236))
        line = np.clip(base + rng.integers(-12, 13, 8), 0, 255).astype(np.uint8)
        if it % 3 == 0:
            line[4:] = np.clip(line[4:].astype(int) + int(rng.integers(-30, 31)), 0, 255)
        bs = int(rng.integers(1, 5))
        ia, ib = int(rng.integers(16, 52)), int(rng.integers(16, 52))
        for chroma in (0, 1):
            buf = line.copy()
            tc0 = S.TC0[bs][ia] if bs < 4 else 0
            fn = L.x264o_deblock_chroma_edge if chroma else L.x264o_deblock_luma_edge
            fn(O.ptr(buf, 4), 1, 8, 1, S.ALPHA[ia], S.BETA[ib], tc0, bs)
            p, q = S.deblock_line([int(line[3 - i]) for i in range(4)], [int(line[4 + i]) for i in range(4)], bs, ia, ib, chroma)
            exp = np.array([p[3], p[2], p[1], p[0], q[0], q[1], q[2], q[3]], np.uint8)
            np.testing.assert_array_equal(buf, exp, err_msg=f"bs {bs} ia {ia} ib {ib} chroma {chroma} line {line}")


def test_deblock_tables_match_spec_copy():
    a = (C.c_uint8 * 52).in_dll(L, "x264o_alpha_table")
    b = (C.c_uint8 * 52).in_dll(L, "x264o_beta_table")
    t = ((C.c_uint8 * 3) * 52).in_dll(L, "x264o_tc0_table")
    assert list(a) == S.ALPHA and list(b) == S.BETA
    for bs in (1, 2, 3):
        assert [t[i][bs - 1] for i in range(52)] == S.TC0[bs]


def test_metric_definitions():
    """SAD/SATD/SA8D against direct matrix definitions (Hadamard via scipy)"""
    from scipy.linalg import hadamard
    rng = np.random.default_rng(11)
    for (w, h) in ((16, 16), (16, 8), (8, 16), (8, 8), (8, 4), (4, 8), (4, 4)):
        a = rng.integers(0, 256, (20, h, w), dtype=np.uint8)
        b = rng.integers(0, 256, (20, h, w), dtype=np.uint8)
        d = a.astype(np.int64) - b
        np.testing.assert_array_equal(O.metric("sad", a, b), np.abs(d).sum((1, 2)))
        H = hadamard(4)
        satd = np.zeros(20, np.int64)
        for y in range(0, h, 4):
            for x in range(0, w, 4):
                satd += np.abs(H @ d[:, y:y + 4, x:x + 4] @ H).sum((1, 2)) >> 1
        np.testing.assert_array_equal(O.metric("satd", a, b), satd)
        if (w, h) in ((8, 8), (16, 16)):
            H8 = hadamard(8)
            raw = np.zeros(20, np.int64)
            for y in range(0, h, 8):
                for x in range(0, w, 8):
                    raw += np.abs(H8 @ d[:, y:y + 8, x:x + 8] @ H8).sum((1, 2))
            np.testing.assert_array_equal(O.metric("sa8d", a, b), (raw + 2) >> 2)


def test_optimize_chroma_2x2_dc_keeps_the_reconstruction():
    """x264o_optimize_chroma_2x2_dc ([x264-upstream] quant.c optimize_chroma_2x2_dc): the trimmed DC levels must dequantise to
    the same per-block DC contribution ((idct2x2 * dmf >> 5) + 32) >> 6 as the original ones, never grow, and return 0 exactly
    when that contribution is zero everywhere."""
    rnd = np.random.default_rng(7)
    dq0 = [160, 176, 208, 224, 256, 288]                  # dequant4_mf[qp % 6][0], flat matrix

    def contrib(d, dmf):
        d = [int(v) for v in d]
        a, b, c, e = d[0] + d[1], d[2] + d[3], d[0] - d[1], d[2] - d[3]
        return [((x * dmf >> 5) + 32) >> 6 for x in (a + b, a - b, c + e, c - e)]
    trimmed = 0
    for _ in range(4000):
        qp = int(rnd.integers(0, 40))
        dmf = dq0[qp % 6] << (qp // 6)
        d = rnd.integers(-6, 7, 4).astype(np.int16) if rnd.random() < 0.8 else rnd.integers(-40, 41, 4).astype(np.int16)
        if not d.any():
            continue
        o = d.copy()
        r = O.L.x264o_optimize_chroma_2x2_dc(O.ptr(o), dmf)
        if dmf > 2048:
            assert r == 1 and np.array_equal(o, d)
            continue
        before = contrib(d, dmf)
        if r == 0:
            assert not any(before), (d, dmf)
        else:
            assert contrib(o, dmf) == before, (d, o, dmf)
            assert np.all(np.abs(o) <= np.abs(d)) and np.all(o * d >= 0)
            trimmed += int(not np.array_equal(o, d))
    assert trimmed > 50

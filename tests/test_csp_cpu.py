"""oracle/csp.c (restatement of /root/reference/csp.c) pinned by closed-form checks: the reference has no tests or
vectors for it and cannot be compiled here (x264vfw.h needs windows.h), so the expected values below come from the
BT.601 / BT.709 definitions and from identities of the conversions, not from running the reference."""
import numpy as np
import pytest

import oracle_lib as O

CSP = O.CSP


def rgb_frame(w, h, bgr, step):
    stride = (3 * w + 3) & ~3 if step == 3 else 4 * w
    buf = np.zeros((h, stride), np.uint8)
    for c in range(3):
        buf[:, c:step * w:step] = bgr[c]
    return buf.reshape(-1)


@pytest.mark.parametrize("mat,full", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_rgb_colour_bars(mat, full):
    """75 % / 100 % bars: Y, Cb, Cr from the matrix definition, within 1 LSB of the 20-bit fixed point"""
    kb, kr = (0.0722, 0.2126) if mat else (0.114, 0.299)
    kg = 1 - kb - kr
    for bgr in [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (191, 191, 0), (16, 128, 235), (7, 99, 201)]:
        b, g, r = bgr
        yl = kr * r + kg * g + kb * b
        if full:
            ey, eu, ev = yl, 128 + 0.5 * (b - yl) / (1 - kb), 128 + 0.5 * (r - yl) / (1 - kr)
        else:
            ey, eu, ev = 16 + 219 * yl / 255, 128 + 224 * 0.5 * (b - yl) / (1 - kb) / 255, 128 + 224 * 0.5 * (r - yl) / (1 - kr) / 255
        for step, csp in ((3, CSP["BGR"]), (4, CSP["BGRA"])):
            out = O.csp_to_i420(rgb_frame(16, 8, bgr, step), csp, 16, 8, mat, full)
            y, u, v = out[:128], out[128:160], out[160:]
            assert len(set(y)) == 1 and len(set(u)) == 1 and len(set(v)) == 1
            assert abs(int(y[0]) - min(ey, 255)) <= 0.75 and abs(int(u[0]) - min(eu, 255)) <= 0.75 and abs(int(v[0]) - min(ev, 255)) <= 0.75, (bgr, y[0], u[0], v[0], ey, eu, ev)
    # studio-range extremes are exact
    out = O.csp_to_i420(rgb_frame(8, 4, (255, 255, 255), 4), CSP["BGRA"], 8, 4, mat, 0)
    assert out[0] == 235 and out[32] == 128 and out[40] == 128
    out = O.csp_to_i420(rgb_frame(8, 4, (0, 0, 0), 4), CSP["BGRA"], 8, 4, mat, 0)
    assert out[0] == 16 and out[32] == 128 and out[40] == 128


def test_rgb_coefficients_sum():
    """luma weights sum to the range scale, chroma weights cancel (grey stays neutral)"""
    import ctypes as C
    for mat in (0, 1):
        for full in (0, 1):
            c = (C.c_uint32 * 12)()
            O.L.x264o_csp_rgb_coefs(mat, full, c)
            scale = (1 << 20) * (1.0 if full else 219 / 255)
            assert abs(c[0] + c[1] + c[2] - scale) <= 2
            assert abs(int(c[6]) - c[4] - c[5]) <= 2 and abs(int(c[8]) - c[9] - c[10]) <= 2


def test_planar_identities():
    rng = np.random.default_rng(3)
    w, h = 24, 12
    i420 = rng.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
    y, u, v = i420[:w * h], i420[w * h:w * h * 5 // 4], i420[w * h * 5 // 4:]
    np.testing.assert_array_equal(O.csp_to_i420(i420, CSP["I420"], w, h), i420)
    np.testing.assert_array_equal(O.csp_to_i420(np.concatenate([y, v, u]), CSP["YV12"], w, h), i420)
    # vertical flip: rows reversed plane by plane
    fl = lambda p, ww: p.reshape(-1, ww)[::-1].reshape(-1)
    np.testing.assert_array_equal(O.csp_to_i420(np.concatenate([fl(y, w), fl(u, w // 2), fl(v, w // 2)]), CSP["I420"] | CSP["VFLIP"], w, h), i420)
    # YV16 / YV24 built by replicating chroma subsample back exactly
    u2, v2 = u.reshape(h // 2, w // 2), v.reshape(h // 2, w // 2)
    yv16 = np.concatenate([y, np.repeat(v2, 2, 0).reshape(-1), np.repeat(u2, 2, 0).reshape(-1)])
    np.testing.assert_array_equal(O.csp_to_i420(yv16, CSP["YV16"], w, h), i420)
    yv24 = np.concatenate([y, np.repeat(np.repeat(v2, 2, 0), 2, 1).reshape(-1), np.repeat(np.repeat(u2, 2, 0), 2, 1).reshape(-1)])
    np.testing.assert_array_equal(O.csp_to_i420(yv24, CSP["YV24"], w, h), i420)
    # rounding of the 2:1 averages: (a+b+1)>>1 and (a+b+c+d+2)>>2
    a = np.zeros((4, 4), np.uint8); a[0, 0], a[1, 0] = 1, 2
    yv16 = np.concatenate([np.zeros(16, np.uint8), a[:, :2].reshape(-1), np.zeros(8, np.uint8)])
    assert O.csp_to_i420(yv16, CSP["YV16"], 4, 4)[16 + 4] == 2          # V plane lands in I420 plane 2
    yv24 = np.concatenate([np.zeros(16, np.uint8), np.array([[1, 2, 0, 0], [0, 0, 0, 0]] * 2, np.uint8).reshape(-1), np.zeros(16, np.uint8)])
    assert O.csp_to_i420(yv24, CSP["YV24"], 4, 4)[16 + 4] == 1          # (1+2+0+0+2)>>2


def test_packed_422():
    rng = np.random.default_rng(4)
    w, h = 16, 6
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    u = rng.integers(0, 256, (h, w // 2), dtype=np.uint8)
    v = rng.integers(0, 256, (h, w // 2), dtype=np.uint8)
    yuyv = np.zeros((h, 2 * w), np.uint8)
    yuyv[:, 0::2] = y; yuyv[:, 1::4] = u; yuyv[:, 3::4] = v
    uyvy = np.zeros((h, 2 * w), np.uint8)
    uyvy[:, 1::2] = y; uyvy[:, 0::4] = u; uyvy[:, 2::4] = v
    eu = ((u[0::2].astype(int) + u[1::2] + 1) >> 1).astype(np.uint8)
    ev = ((v[0::2].astype(int) + v[1::2] + 1) >> 1).astype(np.uint8)
    exp = np.concatenate([y.reshape(-1), eu.reshape(-1), ev.reshape(-1)])
    np.testing.assert_array_equal(O.csp_to_i420(yuyv.reshape(-1), CSP["YUYV"], w, h), exp)
    np.testing.assert_array_equal(O.csp_to_i420(uyvy.reshape(-1), CSP["UYVY"], w, h), exp)
    np.testing.assert_array_equal(O.csp_to_i420(yuyv[::-1].reshape(-1), CSP["YUYV"] | CSP["VFLIP"], w, h), exp)


def test_img_fill_layout():
    """x264vfw_img_fill (codec.c:304-379)"""
    assert O.csp_img_fill(CSP["I420"], 16, 8) == (192, [0, 128, 160], [16, 8, 8])
    assert O.csp_img_fill(CSP["YV16"], 16, 8) == (256, [0, 128, 192], [16, 8, 8])
    assert O.csp_img_fill(CSP["YV24"], 16, 8) == (384, [0, 128, 256], [16, 16, 16])
    assert O.csp_img_fill(CSP["YUYV"], 16, 8)[0] == 256 and O.csp_img_fill(CSP["BGRA"], 16, 8)[0] == 512
    assert O.csp_img_fill(CSP["BGR"], 10, 2) == (64, [0, 0, 0], [32, 0, 0])      # DIB rows are dword aligned
    assert O.csp_img_fill(CSP["NV12"], 16, 8)[0] == -1                           # not an input of the I420 encoder path

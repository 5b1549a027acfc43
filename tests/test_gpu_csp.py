"""HIP colourspace-ingest kernels (x264gpu_csp_to_i420, through the C ABI) == oracle/csp.c, bit-exact, for every input
format the driver feeds an I420 encoder, both flip directions, both matrices/ranges, ragged widths."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
CSP = O.CSP


def gpu_convert(buf, csp, w, h, mat, full):
    import torch
    from x264vfw_amd import lib
    off, st = (C.c_long * 3)(), (C.c_int * 3)()
    n = lib.x264gpu_csp_img_fill(csp, w, h, off, st)
    assert n == buf.size == O.csp_img_fill(csp, w, h)[0]
    d_src = torch.from_numpy(np.ascontiguousarray(buf)).cuda()
    d_dst = torch.full((w * h * 3 // 2,), 0xAA, dtype=torch.uint8, device="cuda")
    src = (C.c_void_p * 3)(*[d_src.data_ptr() + o for o in off])
    dst = (C.c_void_p * 3)(d_dst.data_ptr(), d_dst.data_ptr() + w * h, d_dst.data_ptr() + w * h + (w // 2) * (h // 2))
    lib.check(lib.x264gpu_csp_to_i420(src, st, csp, w, h, mat, full, dst, (C.c_int * 3)(w, w // 2, w // 2), None), "csp_to_i420")
    torch.cuda.synchronize()
    return d_dst.cpu().numpy()


@pytest.mark.parametrize("name", ["I420", "YV12", "YV16", "YV24", "YUYV", "UYVY", "BGR", "BGRA"])
@pytest.mark.parametrize("w,h", [(16, 8), (1920, 1080), (354, 290), (66, 34), (10, 6), (2, 2)])
@pytest.mark.parametrize("flip", [0, 1])
def test_csp_matches_oracle(gpu, name, w, h, flip):
    csp = CSP[name] | (CSP["VFLIP"] if flip else 0)
    n = O.csp_img_fill(csp, w, h)[0]
    rng = np.random.default_rng(w * 31 + h + flip)
    buf = rng.integers(0, 256, n, dtype=np.uint8)
    for mat, full in ([(0, 0)] if name not in ("BGR", "BGRA") else [(0, 0), (0, 1), (1, 0), (1, 1)]):
        np.testing.assert_array_equal(gpu_convert(buf, csp, w, h, mat, full), O.csp_to_i420(buf, csp, w, h, mat, full), err_msg=f"{name} {w}x{h} flip {flip} mat {mat} full {full}")


def test_csp_extreme_values(gpu):
    """saturated inputs exercise the uint32 wrap-free range of the fixed-point sums"""
    for val in (0, 255):
        for name in ("BGR", "BGRA"):
            w, h = 32, 4
            buf = np.full(O.csp_img_fill(CSP[name], w, h)[0], val, np.uint8)
            for mat in (0, 1):
                for full in (0, 1):
                    np.testing.assert_array_equal(gpu_convert(buf, CSP[name], w, h, mat, full), O.csp_to_i420(buf, CSP[name], w, h, mat, full))


def test_csp_rejects_unsupported(gpu):
    from x264vfw_amd import lib
    z = (C.c_void_p * 3)(1, 1, 1)
    s = (C.c_int * 3)(16, 8, 8)
    assert lib.x264gpu_csp_to_i420(z, s, CSP["NV12"], 16, 8, 0, 0, z, s, None) != 0          # another encoder colourspace
    assert lib.x264gpu_csp_to_i420(z, s, CSP["I420"], 15, 8, 0, 0, z, s, None) != 0          # odd width


@pytest.mark.parametrize("name", ["BGRA", "YUYV", "YV24"])
def test_csp_batch(gpu, name):
    """`frames` pictures per launch == the same pictures converted one by one"""
    import torch
    from x264vfw_amd import lib
    w, h, nf = 66, 34, 5
    csp = CSP[name] | CSP["VFLIP"]
    n = O.csp_img_fill(csp, w, h)[0]
    rng = np.random.default_rng(11)
    buf = rng.integers(0, 256, (nf, n), dtype=np.uint8)
    off, st = (C.c_long * 3)(), (C.c_int * 3)()
    lib.x264gpu_csp_img_fill(csp, w, h, off, st)
    d_src = torch.from_numpy(buf).cuda()
    osz = w * h * 3 // 2
    d_dst = torch.zeros((nf, osz), dtype=torch.uint8, device="cuda")
    src = (C.c_void_p * 3)(*[d_src.data_ptr() + o for o in off])
    dst = (C.c_void_p * 3)(d_dst.data_ptr(), d_dst.data_ptr() + w * h, d_dst.data_ptr() + w * h + (w // 2) * (h // 2))
    lib.check(lib.x264gpu_csp_to_i420_batch(src, st, n, csp, w, h, 0, 1, dst, (C.c_int * 3)(w, w // 2, w // 2), osz, nf, None), "csp batch")
    torch.cuda.synchronize()
    got = d_dst.cpu().numpy()
    for i in range(nf):
        np.testing.assert_array_equal(got[i], O.csp_to_i420(buf[i], csp, w, h, 0, 1), err_msg=f"frame {i}")

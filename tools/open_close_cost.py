"""x264_encoder_open / close of N batch sessions on one thread: milliseconds each (under `rocprofv3 --hip-trace --stats` the HIP calls behind them — how the per-session
upload stream was found to cost 0.7 + 0.6 ms).   Usage: open_close_cost.py N"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import host_lib as HL
H = HL.H
n = int(sys.argv[1])
os.environ["X264GPU_BATCH"] = str(n)
def mk():
    p = HL.Param()
    assert H.x264_param_default_preset(C.byref(p), b"medium", None) == 0
    p.i_width, p.i_height, p.i_csp = 1920, 1080, HL.X264_CSP_I420
    p.i_fps_num, p.i_fps_den, p.i_log_level = 25, 1, -1
    for k, v in (("qp", "23"), ("keyint", "250"), ("scenecut", "0"), ("b-adapt", "0"), ("threads", "1")):
        assert H.x264_param_parse(C.byref(p), k.encode(), v.encode()) == 0
    return H.x264_encoder_open_157(C.byref(p))
t0 = time.perf_counter()
hs = [mk() for _ in range(n)]
t1 = time.perf_counter()
for h in hs: H.x264_encoder_close(h)
t2 = time.perf_counter()
print("open %.2f ms each, close %.2f ms each (n=%d, one thread)" % (1e3 * (t1 - t0) / n, 1e3 * (t2 - t1) / n, n))

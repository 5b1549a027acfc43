#!/bin/bash
# The multi-session leg with the batcher's own timers and the caller's per-call trace (profiles/r06_sessions_round_timeline.txt): run through gpurun from the repo root.
#   [SMOOTH=1] bash tools/sessions_timing.sh [sessions] [pictures each]
set -e
mkdir -p gpurun_out/ms
python3 - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import numpy as np
from synth import synth_frames
src = synth_frames(1920, 1080, 16, seed=0x264, scene_len=97)
if os.environ.get("SMOOTH"):          # SMOOTH=1: the textures' per-pixel noise smoothed (7 x 7 box) with a little sensor noise on top: ~58 kB a picture at qp 23 instead of 171
    from scipy.ndimage import uniform_filter
    rng = np.random.default_rng(1)
    out = []
    for fr in src:
        fr = np.ascontiguousarray(fr).copy()
        y = uniform_filter(fr[:1920 * 1080].reshape(1080, 1920).astype(np.float32), 7) + rng.normal(0, 1.5, (1080, 1920))
        fr[:1920 * 1080] = np.clip(y + 0.5, 0, 255).astype(np.uint8).reshape(-1)
        out.append(fr)
    src = out
with open("/dev/shm/ms.yuv", "wb") as f:
    for fr in src: f.write(np.ascontiguousarray(fr).tobytes())
PY
mkdir -p tools/_build
g++ -O2 -std=c++17 -pthread -Iinclude -o tools/_build/multi_session tools/multi_session.cpp -Lx264vfw_amd -lx264gpu_host -Wl,-rpath,'$ORIGIN/../../x264vfw_amd'
MULTI_SESSION_TRACE=1 X264GPU_BATCH_TIMING=1 tools/_build/multi_session /dev/shm/ms.yuv 1920 1080 16 ${1:-2048} ${2:-10} 23 2>&1 | tail -16 | cut -c1-1500
rm -f /dev/shm/ms.yuv

#!/bin/bash
# Kernel trace of the multi-session leg: when every lock-step round ran on the device and how long the device sat idle before it (profiles/r06_sessions_round_timeline.txt).
#   bash tools/sessions_kernel_trace.sh [sessions] [pictures each]      (through gpurun, from the repo root)
set -e
mkdir -p gpurun_out/ms
python3 - <<'PY'
import sys, os
sys.path.insert(0, "tests")
import numpy as np
from synth import synth_frames
src = synth_frames(1920, 1080, 16, seed=0x264, scene_len=97)
with open("/dev/shm/ms.yuv", "wb") as f:
    for fr in src: f.write(np.ascontiguousarray(fr).tobytes())
PY
mkdir -p tools/_build
g++ -O2 -std=c++17 -pthread -Iinclude -o tools/_build/multi_session tools/multi_session.cpp -Lx264vfw_amd -lx264gpu_host -Wl,-rpath,'$ORIGIN/../../x264vfw_amd'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ms/trace -- tools/_build/multi_session /dev/shm/ms.yuv 1920 1080 16 ${1:-2048} ${2:-10} 23 > gpurun_out/ms/run.log 2>&1 || true
tail -2 gpurun_out/ms/run.log | cut -c1-400
f=$(find gpurun_out/ms/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY' > gpurun_out/ms/timeline.txt
import csv, sys, re
rows = list(csv.reader(open(sys.argv[1])))[1:]
rows.sort(key=lambda r: int(r[9]))
t0 = int(rows[0][9])
big = [r for r in rows if int(r[10]) - int(r[9]) > 5e6]
prev_end = None
for r in big:
    m = re.search(r'(k_\w+)', r[7])
    s, e = (int(r[9]) - t0) / 1e9, (int(r[10]) - t0) / 1e9
    print("%-24s q%s st%s start %8.3f end %8.3f dur %7.3f gap_before %7.3f" % (m.group(1) if m else r[7][:24], r[2], r[3], s, e, e - s, s - prev_end if prev_end is not None else 0))
    prev_end = max(prev_end or 0, e)
PY
cat gpurun_out/ms/timeline.txt | head -80
rm -rf gpurun_out/ms/trace /dev/shm/ms.yuv

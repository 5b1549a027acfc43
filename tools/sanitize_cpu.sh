#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU side: the oracle (test infrastructure), the stub behind the device ABI and the product's
# HOST library built from its own sources against that stub — the whole session logic (lookahead decisions, rate control, DPB, entropy coding, muxers)
# runs under the sanitizers; the device code cannot (no GPU sanitizers on this pool).  Builds into a scratch copy, leaves the tree alone.
#   tools/sanitize_cpu.sh [scratch dir]          -> findings on stdout, the full log in <scratch>/log.txt
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=${1:-/tmp/x264gpu_san}
rm -rf "$S" && mkdir -p "$S/repo/x264vfw_amd"
cp -r "$ROOT/oracle" "$ROOT/tests" "$ROOT/include" "$ROOT/tools" "$S/repo/"
cp -r "$ROOT/x264vfw_amd/host" "$ROOT"/x264vfw_amd/*.py "$S/repo/x264vfw_amd/"
for f in "$ROOT"/x264vfw_amd/*.so; do [ -e "$f" ] && cp "$f" "$S/repo/x264vfw_amd/"; done      # (git-ignored: absent on a clean CPU-only checkout; the stub build makes its own)
mkdir -p "$S/repo/x264vfw_amd/csrc" && cp "$ROOT"/x264vfw_amd/csrc/*.inc "$S/repo/x264vfw_amd/csrc/"      # (generated tables a CPU test regenerates and compares)
cd "$S/repo"
rm -f oracle/*.o oracle/liboracle.so x264vfw_amd/host/*.o; rm -rf tests/stub/_build; mkdir -p tests/stub/_build
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
make -s -C oracle liboracle.so CFLAGS="-O1 -g $SAN -ffp-contract=off -mavx2 -mbmi2 -std=c99 -fPIC -D_GNU_SOURCE -I. -I../include" \
     CXXFLAGS="-O1 -g $SAN -std=c++17 -fPIC -I. -I../include" CXX="g++ $SAN"
( cd tests/stub
  gcc -O1 -g $SAN -std=gnu99 -fPIC -shared -Wno-unused-parameter -I../../include -I../../oracle -o _build/libx264gpu.so x264gpu_stub.c -L../../oracle -loracle \
      -Wl,-rpath,'$ORIGIN/../../../oracle' -Wl,-soname,libx264gpu.so
  g++ -O1 -g $SAN -ffp-contract=off -std=c++17 -fPIC -shared -pthread -I../../include -o _build/libx264gpu_host.so ../../x264vfw_amd/host/*.cpp -L_build -lx264gpu \
      -Wl,-rpath,'$ORIGIN' -lm 2>/dev/null )
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0
python -m pytest tests/test_decisions_cpu.py tests/test_cpu_oracle.py tests/test_bframes_cpu.py tests/test_host_cpu.py tests/test_vfw_cpu.py -q -m "not gpu" -p no:cacheprovider -s > "$S/log.txt" 2>&1 || true
# the child sessions' own stderr is swallowed by the tests: a few sessions run directly
for o in "crf=23 rc-lookahead=10 scene_len=14" "crf=22 b-adapt=2 weightp=2 fade=4 rc-lookahead=6" "bitrate=400 rc-lookahead=5 bframes=2" \
         "qp=24 bframes=3 direct=auto trellis=2 subme=9 me=umh ref=3" "crf=25 bframes=0 weightp=0 rc-lookahead=4"; do
    python tests/stub/run_host_b.py "$S/o.h264" 176 144 24 3 $o >> "$S/log.txt" 2>&1 || echo "session failed: $o"
done
# ... and the GOP slots (--threads G) with and without B pictures over several stand-in devices, incl. the stream's last, shorter GOP
for t in 1 3; do
    X264GPU_STUB_DEVICES=2 python tests/stub/run_host.py 96 80 23 7 qp=27 keyint=8 min-keyint=8 scenecut=0 ref=2 bframes=3 b-adapt=0 weightp=2 threads=$t >> "$S/log.txt" 2>&1 || echo "GOP-slot session failed: threads=$t"
    X264GPU_STUB_DEVICES=2 python tests/stub/run_host.py 96 80 14 7 qp=27 keyint=4 min-keyint=4 scenecut=0 ref=2 bframes=0 weightp=0 threads=$t >> "$S/log.txt" 2>&1 || echo "GOP-slot session failed: threads=$t (I / P)"
done
# ... and a session that codes one picture a call beside the default (pictures in flight over launch contexts: the sessions above)
X264GPU_INFLIGHT=0 python tests/stub/run_host_b.py "$S/o.h264" 176 144 24 3 crf=23 rc-lookahead=10 scene_len=14 >> "$S/log.txt" 2>&1 || echo "session failed: one picture a call"
grep -h "passed\|failed" "$S/log.txt" | tail -1
echo "findings:"
grep -h "runtime error\|ERROR: AddressSanitizer" "$S/log.txt" | sed 's/.*\/repo\///' | sort | uniq -c | sort -rn || true

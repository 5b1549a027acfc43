#!/bin/bash
# Where the macroblock loop's instructions are: rocprofv3 PC sampling (beta) of a short bench.py run against a build of the device
# library that carries line tables (make EXTRA=-gline-tables-only in a copy of x264vfw_amd/csrc; same code, X264GPU_LIB points at it).
#   bash tools/pc_sample.sh <tag> <lib.so> [bench.py flags]      (through gpurun, from the repo root)
# Only the histogram (tools/pc_hist.py) is copied back; the raw samples stay in /tmp on the box.
set -u
tag=${1:-pcs}; lib=${2:-}; shift; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pcs_$tag
mkdir -p "$out"
export TMPDIR=/tmp
[ -n "$lib" ] && export X264GPU_LIB=$root/$lib
cd /tmp
rm -rf /tmp/pcs_raw
timeout 900 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit ${PCS_UNIT:-cycles} --pc-sampling-method ${PCS_METHOD:-stochastic} --pc-sampling-interval ${PCS_INTERVAL:-1048576} \
    --kernel-trace --output-format csv -d /tmp/pcs_raw -- python3 $root/bench.py --cpu-frames 0 --e2e-frames 0 "$@" > $out/bench.json 2> $out/rocprof.err
echo "rocprofv3 rc=$?" >> $out/rocprof.err
ls -laR /tmp/pcs_raw | head -40 > $out/files.txt
python3 $root/tools/pc_hist.py /tmp/pcs_raw $out
ls -la $out

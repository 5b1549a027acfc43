#!/bin/bash
# Does any result depend on memory the code never wrote?  Builds the device library twice out of tree with every automatic variable pre-set by the compiler
# (-ftrivial-auto-var-init=zero | pattern: registers and scratch slots that the source leaves uninitialised) and LDS pre-filled with 0xCD at the start of
# the macroblock loop (-DX264GPU_POISON), everything at -O3 — the parity suite must pass on both, i.e. give the checker's results whatever the memory held.
#   bash tools/poison_check.sh build            (here: two libraries into scratch/poison/)
#   bash tools/poison_check.sh run [pytest selection]   (on the GPU box, through gpurun from the repo root)
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
if [ "${1:-}" = "build" ]; then
  mkdir -p $root/scratch/poison
  for init in zero pattern; do
    d=/tmp/poison_$init; rm -rf $d; mkdir -p $d/x264vfw_amd; cp -r $root/include $d/; cp -r $root/x264vfw_amd/csrc $root/x264vfw_amd/host $d/x264vfw_amd/
    ( cd $d/x264vfw_amd/csrc && rm -f *.o && sed -i '/^mb_slice_b_umh.o: OPT/d; /^mb_slice_ref_b_umh.o: OPT/d' Makefile && make -j8 EXTRA="-ftrivial-auto-var-init=$init -DX264GPU_POISON" > $d/build.log 2>&1; echo "$init rc=$?" ) && cp $d/x264vfw_amd/libx264gpu.so $root/scratch/poison/libx264gpu_$init.so
  done
  ls -la $root/scratch/poison
else
  shift || true
  sel=${*:-tests/test_gpu_bframes.py tests/test_gpu_pipeline.py}
  out=$root/gpurun_out/poison; mkdir -p $out; cd $root
  for init in zero pattern; do
    echo "== auto variables = $init, LDS = 0xCD"
    X264GPU_LIB=$root/scratch/poison/libx264gpu_$init.so timeout 3000 python3 -m pytest $sel -q -m gpu > $out/$init.log 2>&1; tail -6 $out/$init.log | cut -c1-300
  done
fi

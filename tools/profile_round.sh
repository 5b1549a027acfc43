#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01 [extra bench.py flags]
# pass 1: --kernel-trace --stats of the default bench workload (average launch duration per kernel)
# pass 2/3: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they cannot share one, MI355X_MICROARCH.md)
# pass 4: SQ issue counters of the pipeline kernels
# Raw output goes to /tmp (it exceeds the gpurun_out size cap); only the summaries are copied back.
set -u
tag=${1:-r01}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
K="--cpu-frames 0 --e2e-frames 0 --lookahead 0 --survey-leg 0"
P="--cpu-frames 0 --e2e-frames 0 --lookahead 0 --survey-leg 0"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kt -- python3 $root/bench.py $K "$@" > $out/bench_under_rocprof.json 2> /tmp/p_kt.err
cp $(ls /tmp/p_kt/*/*kernel_stats.csv | head -1) $out/kernel_stats_full.csv
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "x264gpu" --output-format csv -d /tmp/p_rd -- python3 $root/bench.py $P "$@" > /tmp/p_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "x264gpu" --output-format csv -d /tmp/p_wr -- python3 $root/bench.py $P "$@" > /tmp/p_wr.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex "x264gpu" --output-format csv -d /tmp/p_sq -- python3 $root/bench.py $P "$@" > /tmp/p_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-include-regex "x264gpu" --output-format csv -d /tmp/p_sq2 -- python3 $root/bench.py $P "$@" > /tmp/p_sq2.log 2>&1
# pass 6: lane utilisation of the macroblock loop: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) per instantiation
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "k_mb_slice" --output-format csv -d /tmp/p_lane -- python3 $root/bench.py $P --lookahead 0 "$@" > /tmp/p_lane.log 2>&1
python3 - "$(ls /tmp/p_lane/*/*counter_collection.csv 2>/dev/null | head -1)" > $out/pmc_lane_utilisation.txt <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'][:70]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k in acc:
    print(k, 'launches', len(n[k]))
    for c,v in sorted(acc[k].items()): print('   %-32s %.5g per launch' % (c, v/len(n[k])))
    a=acc[k]
    if a.get('SQ_ACTIVE_INST_VALU'): print('   lane utilisation (THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU)) = %.3f' % (a['SQ_THREAD_CYCLES_VALU']/(64*a['SQ_ACTIVE_INST_VALU'])))
PY
python3 $root/tools/profile_summarise.py $out /tmp/p_kt /tmp/p_rd /tmp/p_wr /tmp/p_sq /tmp/p_sq2
ls -la $out

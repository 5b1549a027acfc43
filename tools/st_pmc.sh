#!/bin/bash
# PMC counters of the lookahead's frame-cost kernel (k_st_cost) for tools/st_time.py's launches:  bash tools/st_pmc.sh <tag> [streams]   (through gpurun, from the repo root)
set -u
tag=${1:-st}; S=${2:-256}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/stp1 /tmp/stp2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES --kernel-include-regex "k_st_cost" --output-format csv -d /tmp/stp1 -- python3 $root/tools/st_time.py $S > $out/st_time_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-include-regex "k_st_cost" --output-format csv -d /tmp/stp2 -- python3 $root/tools/st_time.py $S > $out/st_time_pmc2.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum --kernel-include-regex "k_st_cost" --output-format csv -d /tmp/stp3 -- python3 $root/tools/st_time.py $S > $out/st_time_pmc3.log 2>&1
python3 - $S > $out/lookahead_pmc.txt <<'PY'
import csv, glob, sys, collections
S = int(sys.argv[1])
acc = collections.Counter(); n = set()
for d in ("/tmp/stp1", "/tmp/stp2", "/tmp/stp3"):
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add((d, r["Dispatch_Id"]))
launches = len({x for x in n if x[0] == "/tmp/stp1"})
blocks = launches * S * 8160
print("k_st_cost: %d launches x %d streams x 8160 blocks = %.2f M blocks" % (launches, S, blocks / 1e6))
for k, v in sorted(acc.items()): print("  %-30s %12.4g   = %9.2f a block" % (k, v, v / blocks))
if acc.get("SQ_WAVE_CYCLES"): print("  SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f   SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES"], acc["SQ_ACTIVE_INST_ANY"] / acc["SQ_WAVE_CYCLES"]))
PY
cat $out/lookahead_pmc.txt; tail -2 $out/st_time_pmc1.log

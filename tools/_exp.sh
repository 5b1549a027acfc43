set -u
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_b1 -- python3 $GRAFT_REPO_ROOT/tools/bench_b1.py 24 1 250 qp slices=68 > $GRAFT_REPO_ROOT/gpurun_out/b1_slices.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/b1_slices.log
head -14 $(ls /tmp/p_b1/*/*kernel_stats.csv | head -1) | cut -c1-150

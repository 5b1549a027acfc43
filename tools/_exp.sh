set -u
cd $GRAFT_REPO_ROOT
timeout 300 python tools/parity_mb.py > gpurun_out/parity.log 2>&1; tail -1 gpurun_out/parity.log
B="--cpu-frames 0 --e2e-frames 0 --steps 6 --warmup 2"
python bench.py $B --streams 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('idr_ms_per_step'))"
make -s -C x264vfw_amd/csrc clean; make -s -C x264vfw_amd/csrc -j16 "EXTRA=-DMB_PROF" 2>&1 | grep -E "error" | head
python tools/mb_prof.py 2048 4 2>&1 | tail -3

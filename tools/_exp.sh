set -u
cd $GRAFT_REPO_ROOT
timeout 300 python tools/parity_mb.py > gpurun_out/parity.log 2>&1; tail -1 gpurun_out/parity.log
timeout 300 python -m pytest tests/test_gpu_prims.py -q -x 2>&1 | tail -2
B="--cpu-frames 0 --e2e-frames 0 --steps 6 --warmup 2"
python bench.py $B --streams 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_ms'], d['roofline'].get('idr_ms_per_step'))"

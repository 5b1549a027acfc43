set -u
cd $GRAFT_REPO_ROOT
make -s -C x264vfw_amd/csrc clean; make -s -C x264vfw_amd/csrc -j16 "EXTRA=-DMB_PROF" 2>&1 | grep -E "error" | head
python tools/mb_prof.py 2048 4 2>&1 | tail -9 > gpurun_out/mb_prof.log
make -s -C x264vfw_amd/csrc clean; make -s -C x264vfw_amd/csrc -j16 "EXTRA=-DMB_PROF -DMB_PROF_RD" 2>&1 | grep -E "error" | head
python tools/mb_prof.py 2048 4 2>&1 | tail -9 >> gpurun_out/mb_prof.log
cat gpurun_out/mb_prof.log

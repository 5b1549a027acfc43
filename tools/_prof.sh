cd $GRAFT_REPO_ROOT
make -s -C x264vfw_amd/csrc clean >/dev/null 2>&1
make -s -C x264vfw_amd/csrc -j32 EXTRA="-DMB_PROF -DMB_PROF_RD" 2>&1 | grep -E "error" | head
python tools/mb_prof.py 2048 9 2>&1 | tail -22

cd $GRAFT_REPO_ROOT
timeout 1500 python tools/fuzz_reductions.py 600 91 2>&1 | tail -12

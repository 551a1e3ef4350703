cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default.log 2>&1
grep -v amdgpu.ids gpurun_out/bench_default.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

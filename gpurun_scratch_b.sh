cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/bench_default.log 2>&1
tail -5 gpurun_out/bench_default.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o d24_b64 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_r01.log 2>&1
ls -la gpurun_out/prof_r01/
rm -f gpurun_out/prof_r01/*kernel_trace.csv

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python tools/fuzz_gemm.py 1200 41 2>&1 | tail -3
python tools/parity_report.py --run > gpurun_out/parity_run.log 2>&1; grep -n "passed\|failed" gpurun_out/parity_run.log | tail -2; tail -1 gpurun_out/parity_run.log
for i in 1 2; do timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-extras 2>&1 | tail -1 | cut -c1-220; done

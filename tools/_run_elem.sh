cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train_kernels.py tests/test_torch_ops_slots.py -x -q -m gpu > gpurun_out/tn_tests.log 2>&1; tail -3 gpurun_out/tn_tests.log
timeout 900 python tools/fuzz_gemm.py 600 77 2>&1 | tail -2
timeout 600 python tools/train_bench.py 30 8 2>&1 | tail -3

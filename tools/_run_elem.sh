cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_kernels.py -x -q -m gpu -k "gemm or split" > gpurun_out/sk_tests.log 2>&1; tail -2 gpurun_out/sk_tests.log
timeout 600 python tools/_lat.py 2>&1 | grep -v amdgpu.ids | tail -3
for i in 1 2; do timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-extras 2>&1 | tail -1 | cut -c1-170; done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_torch_ops_slots.py -x -q -m gpu > gpurun_out/train_tests.log 2>&1; tail -5 gpurun_out/train_tests.log
timeout 300 python tools/elem_bench.py > gpurun_out/elem_bench.log 2>&1; grep -v amdgpu.ids gpurun_out/elem_bench.log
for i in 1 2; do timeout 600 python bench.py --mode train --steps 6 --warmup 2 --no-extras 2>&1 | tail -1; done

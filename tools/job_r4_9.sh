mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" > gpurun_out/attn_tests.log 2>&1; tail -5 gpurun_out/attn_tests.log
timeout 900 python tools/fuzz_attn.py 600 21 > gpurun_out/fuzz_attn_pipe.log 2>&1; tail -3 gpurun_out/fuzz_attn_pipe.log
echo "== pipe" > gpurun_out/attn_bench_pipe.txt; timeout 600 python tools/attn_bench.py 384 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn_bench_pipe.txt
echo "== round-3 kernel (CVAR_ATTN_PIPE=0)" >> gpurun_out/attn_bench_pipe.txt; CVAR_LIB=ab/libcvar_nopipe.so timeout 600 python tools/attn_bench.py 384 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/attn_bench_pipe.txt
cat gpurun_out/attn_bench_pipe.txt

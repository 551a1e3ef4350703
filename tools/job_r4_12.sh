mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or conv or linear or split" > gpurun_out/gemm_tests_sk.log 2>&1; tail -6 gpurun_out/gemm_tests_sk.log
timeout 900 python tools/fuzz_gemm.py 600 31 > gpurun_out/fuzz_gemm_sk.log 2>&1; tail -2 gpurun_out/fuzz_gemm_sk.log
for cfg in 0 9; do echo "== tile_cfg $cfg (9 = two-kernel split-K)"; ISO_CFG=$cfg LAT_B=1,4,8,16 python tools/latency_bench.py 24 2>&1 | tail -1; done | tee gpurun_out/lat_sk.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/parity_sk.log 2>&1; tail -3 gpurun_out/parity_sk.log

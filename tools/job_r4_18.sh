mkdir -p gpurun_out
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "d24 or large_batch" -s > gpurun_out/new_tests.log 2>&1; grep -E "passed|failed|\[bf16\] d24|\[parity\]|Error|assert" gpurun_out/new_tests.log | head -20

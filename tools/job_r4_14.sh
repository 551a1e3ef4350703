mkdir -p gpurun_out; rm -f gpurun_out/iso_epi.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or linear or split or persistent" > gpurun_out/gemm_tests_epi.log 2>&1; tail -3 gpurun_out/gemm_tests_epi.log
timeout 900 python tools/fuzz_gemm.py 400 41 > gpurun_out/fuzz_gemm_epi.log 2>&1; tail -2 gpurun_out/fuzz_gemm_epi.log
for lib in "" ab/libcvar_noepipe.so; do for cfg in 0 3; do
  echo "== lib ${lib:-epi_pipe} cfg $cfg" >> gpurun_out/iso_epi.txt
  CVAR_LIB=$lib ISO_CFG=$cfg timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_epi.txt
  CVAR_LIB=$lib ISO_CFG=$cfg ISO_EPI=1 timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_epi.txt
done; done
cat gpurun_out/iso_epi.txt

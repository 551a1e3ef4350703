set -x
mkdir -p gpurun_out
CVAR_LIB=ab/libcvar_xb2.so python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or conv or linear" > gpurun_out/gemm_tests_xb2.log 2>&1; tail -3 gpurun_out/gemm_tests_xb2.log
CVAR_LIB=ab/libcvar_xb2.so python tools/fuzz_gemm.py 300 12 > gpurun_out/fuzz_gemm_xb2.log 2>&1; tail -2 gpurun_out/fuzz_gemm_xb2.log
for cfg in 0 3; do for lib in "" ab/libcvar_xb2.so; do
  echo "== cfg $cfg lib ${lib:-xb1}" >> gpurun_out/iso_xb2.txt
  CVAR_LIB=$lib ISO_CFG=$cfg python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_xb2.txt
  CVAR_LIB=$lib ISO_CFG=$cfg ISO_DATA=zeros python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_xb2.txt
done; done
cat gpurun_out/iso_xb2.txt
for cfg in 0 3; do for shape in "4608 1536" "1536 1536" "1536 6144"; do
  echo "== timeline cfg $cfg $shape" >> gpurun_out/timeline.txt
  CVAR_LIB=ab/libcvar_timing.so ISO_CFG=$cfg python tools/gemm_wg_timeline.py $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/timeline.txt
done; done
CVAR_LIB=ab/libcvar_timing.so ISO_DATA=zeros python tools/gemm_wg_timeline.py 4608 1536 2>&1 | grep -v amdgpu.ids >> gpurun_out/timeline.txt
cat gpurun_out/timeline.txt

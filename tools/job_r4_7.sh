mkdir -p gpurun_out; rm -f gpurun_out/timeline_pers.txt gpurun_out/iso_pers2.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent" > gpurun_out/pers_tests.log 2>&1; tail -3 gpurun_out/pers_tests.log
for cfg in 0 7; do
  echo "== cfg $cfg" >> gpurun_out/iso_pers2.txt
  ISO_CFG=$cfg timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_pers2.txt
  ISO_CFG=$cfg ISO_DATA=zeros timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_pers2.txt
done
cat gpurun_out/iso_pers2.txt
for cfg in 0; do for shape in "4608 1536"; do
  echo "== timeline cfg $cfg $shape" >> gpurun_out/timeline_pers.txt
  CVAR_LIB=ab/libcvar_timing.so ISO_CFG=$cfg python tools/gemm_wg_timeline.py $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/timeline_pers.txt
done; done
cat gpurun_out/timeline_pers.txt

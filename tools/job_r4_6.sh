mkdir -p gpurun_out; rm -f gpurun_out/timeline_pers.txt
for cfg in 0 7; do for shape in "4608 1536" "1536 6144"; do
  echo "== timeline cfg $cfg $shape" >> gpurun_out/timeline_pers.txt
  CVAR_LIB=ab/libcvar_timing.so ISO_CFG=$cfg python tools/gemm_wg_timeline.py $shape 2>&1 | grep -v amdgpu.ids >> gpurun_out/timeline_pers.txt
done; done
cat gpurun_out/timeline_pers.txt

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or linear" > gpurun_out/imm_tests.log 2>&1; tail -3 gpurun_out/imm_tests.log
if grep -q "failed" gpurun_out/imm_tests.log; then exit 0; fi
for v in default immoff default immoff; do
  if [ $v = default ]; then unset CVAR_LIB; else export CVAR_LIB=ab/libcvar_$v.so; fi
  timeout 300 python tools/gemm_iso.py 2>&1 | grep -v amdgpu.ids
done
for v in default immoff; do
  if [ $v = default ]; then unset CVAR_LIB; else export CVAR_LIB=ab/libcvar_$v.so; fi
  echo "== $v"
  timeout 600 python tools/gemm_insitu.py 24 128 2>&1 | grep -v amdgpu.ids | head -8 | cut -c1-150
done

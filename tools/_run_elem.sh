cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CVAR_LIB=ab/libcvar_ppair.so timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or linear" > gpurun_out/pp_tests.log 2>&1; tail -2 gpurun_out/pp_tests.log
for v in default ppair ppair0 default ppair; do
  if [ $v = default ]; then unset CVAR_LIB; else export CVAR_LIB=ab/libcvar_$v.so; fi
  echo "== $v"
  timeout 600 python tools/gemm_insitu.py 24 128 2>&1 | grep -v amdgpu.ids | head -6 | cut -c1-150
done

R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3
timeout 400 python tools/fuzz_attn.py 500 78 2>&1 | tail -2
for v in noq64 base noq64 base; do
  echo "=== $v"
  if [ $v = base ]; then L=$R/controlvar_amd/libcvar_hip.so; else L=$R/ab/libcvar_$v.so; fi
  CVAR_LIB=$L timeout 200 python tools/attn_bench.py 128 2 2>&1 | grep -E "scale [6-9]|all scales" | sed 's/v1 .*v2p/v2p/'
done > $O/s5_attn_q64.txt 2>&1
cat $O/s5_attn_q64.txt

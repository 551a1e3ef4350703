R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv or halo or sweep_inside" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "same_bits or batch_rows" 2>&1 | tail -2
for v in narrow base narrow base; do
  echo "=== $v"
  if [ $v = base ]; then L=$R/controlvar_amd/libcvar_hip.so; else L=$R/ab/libcvar_$v.so; fi
  CVAR_LIB=$L timeout 200 python tools/conv_halo_ab.py 5 2>&1 | grep "res=" | sed 's/| implicit.*//'
done > $O/s5_conv_wide.txt 2>&1
cat $O/s5_conv_wide.txt
for v in narrow base; do if [ $v = base ]; then L=$R/controlvar_amd/libcvar_hip.so; else L=$R/ab/libcvar_$v.so; fi; echo "== vae_bench $v"; CVAR_LIB=$L timeout 200 python tools/vae_bench.py 2>&1 | tail -1; done

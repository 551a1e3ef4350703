set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or conv or linear" > gpurun_out/gemm_tests.log 2>&1; tail -3 gpurun_out/gemm_tests.log
python tools/fuzz_gemm.py 400 11 > gpurun_out/fuzz_gemm_xb.log 2>&1; tail -3 gpurun_out/fuzz_gemm_xb.log
for cfg in 0 3; do for lib in "" ab/libcvar_noxb.so; do
  echo "== cfg $cfg lib ${lib:-xb}" >> gpurun_out/iso_xb.txt
  CVAR_LIB=$lib ISO_CFG=$cfg python tools/gemm_iso.py 131072 10 >> gpurun_out/iso_xb.txt 2>&1
  CVAR_LIB=$lib ISO_CFG=$cfg ISO_EPI=1 python tools/gemm_iso.py 131072 10 >> gpurun_out/iso_xb.txt 2>&1
done; done
for lib in "" ab/libcvar_noxb.so; do echo "== zeros cfg 0 lib ${lib:-xb}" >> gpurun_out/iso_xb.txt; CVAR_LIB=$lib ISO_DATA=zeros python tools/gemm_iso.py 131072 10 >> gpurun_out/iso_xb.txt 2>&1; done
cat gpurun_out/iso_xb.txt
python bench.py --steps 3 --warmup 1 > gpurun_out/bench_xb.json 2> gpurun_out/bench_xb.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_xb.json')); print(d['value'], d['roofline']['achieved'], {k:v['value'] for k,v in d['side_configs'].items()})
PY
python -m pytest tests/test_gpu_train.py tests/test_gpu_train_kernels.py tests/test_gpu_variants.py -m gpu -x -q > gpurun_out/gputest2.log 2>&1; tail -5 gpurun_out/gputest2.log

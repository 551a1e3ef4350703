mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "persistent" > gpurun_out/pers_tests.log 2>&1; tail -15 gpurun_out/pers_tests.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "gemm or conv or linear" > gpurun_out/gemm_tests_pers.log 2>&1; tail -3 gpurun_out/gemm_tests_pers.log
rm -f gpurun_out/iso_pers.txt
for cfg in 0 7 3 8; do
  echo "== cfg $cfg" >> gpurun_out/iso_pers.txt
  ISO_CFG=$cfg timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_pers.txt
  ISO_CFG=$cfg ISO_EPI=1 timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_pers.txt
done
echo "== zeros cfg 0" >> gpurun_out/iso_pers.txt; ISO_DATA=zeros timeout 300 python tools/gemm_iso.py 131072 10 2>&1 | grep -v amdgpu.ids >> gpurun_out/iso_pers.txt
cat gpurun_out/iso_pers.txt
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_pers.json 2> gpurun_out/bench_pers.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_pers.json')); print(d['value'], d['roofline']['achieved'], {k:v['value'] for k,v in d['side_configs'].items()})
PY

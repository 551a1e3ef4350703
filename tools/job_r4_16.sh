mkdir -p gpurun_out; rm -f gpurun_out/batch_sweep.txt
for b in 384 512 576; do
  python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --batch $b > gpurun_out/bb.json 2> gpurun_out/bb.err
  python - $b <<'PY' >> gpurun_out/batch_sweep.txt
import json, sys
try:
    d = json.load(open('gpurun_out/bb.json')); print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['achieved'], d['peak_hbm_gb'])
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/bb.err').read()[-400:])
PY
done
cat gpurun_out/batch_sweep.txt

mkdir -p gpurun_out; rm -f gpurun_out/parity_report.jsonl
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_final.log 2>&1; grep -E "passed|failed" gpurun_out/gputest_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_final.json'))
print(d['value'], d['ms_per_step'], d['config']['batch_per_gpu'], d['roofline']['achieved'], d['roofline']['frac'], d['peak_hbm_gb'])
print({k:(v.get('value'), v.get('ms_per_step')) for k,v in d['side_configs'].items()} if isinstance(d.get('side_configs'), dict) else d.get('side_configs'))
print(d['cpu_baseline']['value'], d.get('rccl_ranks'), d.get('per_rank_value'))
PY
python tools/parity_report.py > /dev/null 2>&1; ls gpurun_out | head -30

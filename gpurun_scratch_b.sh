cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout=120 2>&1 | tail -3
python bench.py --depth 24 --batch 64 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d24 -o d24 -- python bench.py --depth 24 --batch 64 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_d24.log 2>&1
python - <<'PY'
import sqlite3
con = sqlite3.connect('gpurun_out/prof_d24/d24_results.db')
cur = con.cursor()
for r in cur.execute("select * from top_kernels limit 12"):
    print(tuple((x[:60] if isinstance(x,str) else (round(x,3) if isinstance(x,float) else x)) for x in r))
PY
rm -f gpurun_out/prof_d24/*.db

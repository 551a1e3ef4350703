mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b1 --output-format csv -- python $R/tools/b1_profile.py 1 > $R/gpurun_out/prof_b1.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b8 --output-format csv -- python $R/tools/b1_profile.py 8 > $R/gpurun_out/prof_b8.log 2>&1
cd $R
for t in b1 b8; do f=$(find gpurun_out/prof_$t -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${t}_kernel_stats.csv; f=$(find gpurun_out/prof_$t -name "*kernel_trace.csv" | head -1); python - "$f" $t <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last generation only: take the last quarter of the launches
n = len(rows) // 4
last = rows[-n:]
t0, t1 = int(last[0]['Start_Timestamp']), int(last[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in last)
print(f'{sys.argv[2]}: {n} launches per generation, span {(t1 - t0) / 1e6:.2f} ms, kernel-busy {busy / 1e6:.2f} ms, mean kernel {busy / n / 1e3:.2f} us, mean gap {(t1 - t0 - busy) / n / 1e3:.2f} us')
acc = collections.defaultdict(lambda: [0, 0])
for r in last:
    k = r['Kernel_Name'].split('(')[0][:70]
    acc[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); acc[k][1] += 1
for k, (ns, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f'   {ns / 1e6:7.3f} ms  {c:5d} x {ns / c / 1e3:7.2f} us  {k}')
PY
done
rm -rf gpurun_out/prof_b1 gpurun_out/prof_b8
python tools/latency_bench.py 24 2>&1 | tail -1

mkdir -p gpurun_out
python tools/gemm_group_sweep.py 196608 10 1,2,4,8,16 2>&1 | grep -v amdgpu.ids > gpurun_out/group_sweep.txt; cat gpurun_out/group_sweep.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for gm in 2 4 8; do
  rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_gm$gm --output-format csv -- python $R/tools/gemm_group_sweep.py 196608 1 $gm > $R/gpurun_out/pmc_gm$gm.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob
for gm in (2, 4, 8):
    f = glob.glob(f'gpurun_out/pmc_gm{gm}/**/*counter_collection.csv', recursive=True)
    if not f: print(gm, 'no csv'); continue
    rows = [r for r in csv.DictReader(open(f[0])) if 'cvar_gemm_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
    # launches in order: per shape (qkv, fc1, fc2, proj): warm launch + timed launch
    vals = [float(r['Counter_Value']) * 1024 * 2 / 1e9 for r in rows]
    print(f'GM={gm}: FETCH_SIZE x2 (GB) per launch, in launch order: ' + ' '.join(f'{v:.2f}' for v in vals))
PY
rm -rf gpurun_out/pmc_gm*/
python -m pytest tests -m gpu -x -q > gpurun_out/gputest_full.log 2>&1; tail -4 gpurun_out/gputest_full.log

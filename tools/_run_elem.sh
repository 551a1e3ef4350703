R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
rocprofv3 --kernel-trace --stats -d $O/tprof --output-format csv -- python $R/bench.py --mode train --steps 3 --warmup 1 --no-extras > $O/tprof.log 2>&1
cd $R
f=$(find gpurun_out/tprof -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/train_kernel_stats_now.csv
rm -rf gpurun_out/tprof
grep -o '"value": [0-9.]*' gpurun_out/tprof.log | head -1

R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/s5_gputest7.log 2>&1; echo "pytest rc=$?" >> $O/s5_gputest7.log
tail -3 $O/s5_gputest7.log; grep "^FAILED" $O/s5_gputest7.log
timeout 300 python tools/vae_bench.py 2>&1 | tail -8
SMALLM=1 LAT_B=1 timeout 300 python tools/latency_bench.py 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/s5_vae_prof --output-format csv -- python $R/tools/vae_bench.py > $O/s5_vae_prof.log 2>&1
f=$(find $O/s5_vae_prof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -c1-150; cp $f $O/s5_vae_kernel_stats.csv; rm -rf $O/s5_vae_prof

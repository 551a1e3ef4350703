R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x > $O/s5_gputest6.log 2>&1; echo "pytest rc=$?" >> $O/s5_gputest6.log
tail -3 $O/s5_gputest6.log; grep "^FAILED" $O/s5_gputest6.log
SMALLM=1 LAT_B=1,2,4,8,16,32 timeout 300 python tools/latency_bench.py 2>/dev/null | tee $O/s5_lat4.json
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
  timeout 300 rocprofv3 --kernel-trace -d $O/s5_tr5_b$B --output-format csv -- python $R/tools/b1_profile.py $B > $O/s5_tr5_b$B.log 2>&1
  f=$(find $O/s5_tr5_b$B -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_by_scale.py $f --decode > $O/s5_scale5_b$B.txt 2>&1
  rm -rf $O/s5_tr5_b$B
done

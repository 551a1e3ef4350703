R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/s5_gputest5.log 2>&1; echo "pytest rc=$?" >> $O/s5_gputest5.log
tail -3 $O/s5_gputest5.log; grep "^FAILED" $O/s5_gputest5.log
timeout 400 python tools/skinny_bench.py 64 100 144 256 400 512 676 800 1024 > $O/s5_skinny_bench3.txt 2>&1; grep "M=" $O/s5_skinny_bench3.txt
SMALLM=1 LAT_B=1,2,4,8,16,32 timeout 300 python tools/latency_bench.py 2>/dev/null | tee $O/s5_lat3.json

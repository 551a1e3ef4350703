R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "skinny or adaln_of_the_next or split_k" > $O/s5_t1.log 2>&1; echo "rc=$?" >> $O/s5_t1.log
tail -4 $O/s5_t1.log
timeout 600 python tools/skinny_bench.py > $O/s5_skinny_bench.txt 2>&1; cat $O/s5_skinny_bench.txt | tail -12
timeout 900 python -m pytest tests -m gpu -x -q > $O/s5_gputest2.log 2>&1; echo "pytest rc=$?" >> $O/s5_gputest2.log
tail -5 $O/s5_gputest2.log
for sm in 1 0; do SMALLM=$sm LAT_B=1,2,4,8 timeout 300 python tools/latency_bench.py > $O/s5_lat_sm$sm.json 2> $O/s5_lat_sm$sm.err; echo "smallm $sm"; cat $O/s5_lat_sm$sm.json; done

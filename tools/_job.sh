R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python tools/parity_report.py --run > $O/s5_parity.log 2>&1; tail -3 $O/s5_parity.log | cut -c1-400
(timeout 900 python tools/fuzz_gemm.py 1200 41 2>&1 | tail -3; timeout 400 python tools/fuzz_attn.py 300 42 2>&1 | tail -2) > $O/r04_fuzz_final.txt 2>&1; cat $O/r04_fuzz_final.txt
SMALLM=1 LAT_B=1,8 timeout 300 python tools/latency_bench.py 2>/dev/null

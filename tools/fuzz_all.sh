cd $GRAFT_REPO_ROOT
(python tools/fuzz_gemm.py 1500 31 2>&1 | tail -2; python tools/fuzz_attn.py 600 32 2>&1 | tail -2; python tools/fuzz_attn_bwd.py 250 33 2>&1 | tail -2; python tools/fuzz_norms.py 400 34 2>&1 | tail -2; python tools/fuzz_reductions.py 500 35 2>&1 | tail -2; python tools/fuzz_resample.py 2>&1 | tail -2) > gpurun_out/r06_fuzz_final.txt 2>&1
cat gpurun_out/r06_fuzz_final.txt

#!/bin/bash
# Round-end measurement set on one MI355X box (run through gpurun): bench line (with board telemetry), rocprofv3 kernel stats of the same command, PMC passes
# (matrix-pipe utilisation; fabric traffic of the GEMM family - FETCH_SIZE and WRITE_SIZE in SEPARATE passes), training profile, the two-rank training line on the
# shared GPU, the GPU suite with its parity records.  tools/current_md.py <tag> turns the copies under profiles/ into profiles/CURRENT.md.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
BENCH="python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-probe --no-telemetry"
python $R/bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof --output-format csv -- $BENCH > $O/${TAG}_prof.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma --output-format csv -- $BENCH > $O/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- $BENCH > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_train --output-format csv -- python $R/bench.py --mode train --steps 3 --warmup 1 > $O/${TAG}_prof_train.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof_vae --output-format csv -- python $R/tools/vae_bench.py > $O/${TAG}_prof_vae.log 2>&1
cd $R
timeout 600 python bench.py --mode train --gpus 2 --share-gpu --train-batch 8 --steps 3 --warmup 1 > $O/${TAG}_train_share_gpu_line.json 2> $O/${TAG}_train_share_gpu.err
python tools/pmc_mfma_util.py gpurun_out/pmc_mfma gpurun_out/${TAG}_pmc_mfma_util.json > /dev/null
python tools/pmc_traffic.py "one d24 B=512 generation x 3 (bench.py --steps 2 --warmup 1), round ${TAG}" > /dev/null
cp gpurun_out/gemm_hbm_traffic.json gpurun_out/${TAG}_gemm_hbm_traffic.json
for d in ${TAG}_prof ${TAG}_prof_train ${TAG}_prof_vae; do f=$(find gpurun_out/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${d}_kernel_stats.csv; done
f=$(find gpurun_out/${TAG}_prof -name "*agent_info.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_agent_info.csv
# keep the raw dirs small: drop them
rm -rf gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_train gpurun_out/${TAG}_prof_vae gpurun_out/pmc_mfma gpurun_out/pmc_fetch gpurun_out/pmc_write
python tools/parity_report.py --run --out gpurun_out/${TAG}_parity_report.json > $O/${TAG}_parity_run.log 2>&1
tail -c 1500 gpurun_out/${TAG}_bench_default.json; echo; head -8 gpurun_out/${TAG}_prof_kernel_stats.csv | cut -c1-160; cat gpurun_out/${TAG}_pmc_mfma_util.json | head -40; cat gpurun_out/${TAG}_gemm_hbm_traffic.json; tail -2 gpurun_out/${TAG}_prof_train.log | cut -c1-800; tail -3 $O/${TAG}_parity_run.log; tail -c 600 $O/${TAG}_train_share_gpu_line.json

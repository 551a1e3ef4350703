#!/bin/bash
# copy the files tools/final_measure.sh <tag> left under gpurun_out/ to profiles/ under their published names, then regenerate profiles/CURRENT.md and README's table
TAG=${1:-r06}; cd "$(dirname "$0")/.."
G=gpurun_out; P=profiles
cp $G/${TAG}_bench_default.json $P/${TAG}_bench_default_line.json
cp $G/${TAG}_prof_kernel_stats.csv $P/${TAG}_d24_b512_kernel_stats.csv
cp $G/${TAG}_prof_train_kernel_stats.csv $P/${TAG}_train_d24_b32_kernel_stats.csv
cp $G/${TAG}_prof_vae_kernel_stats.csv $P/${TAG}_vae_b128_kernel_stats.csv
cp $G/${TAG}_pmc_mfma_util.json $G/${TAG}_gemm_hbm_traffic.json $G/${TAG}_parity_report.json $G/${TAG}_agent_info.csv $G/${TAG}_train_share_gpu_line.json $P/
cp $G/${TAG}_gemm_hbm_traffic.json $P/gemm_hbm_traffic.json
python tools/current_md.py $TAG && python tools/fill_readme.py $TAG

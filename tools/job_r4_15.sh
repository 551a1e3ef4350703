mkdir -p gpurun_out
bash tools/final_measure.sh r04 > gpurun_out/final_measure_r04.log 2>&1
tail -30 gpurun_out/final_measure_r04.log | cut -c1-400
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_vae --output-format csv -- python $R/tools/vae_bench.py 128 > $R/gpurun_out/prof_vae.log 2>&1
cd $R; f=$(find gpurun_out/prof_vae -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_vae_b128_kernel_stats.csv; rm -rf gpurun_out/prof_vae
head -25 gpurun_out/r04_vae_b128_kernel_stats.csv | cut -c1-150

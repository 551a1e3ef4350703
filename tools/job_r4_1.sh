set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gputest.log 2>&1; echo "pytest rc $?" >> gpurun_out/gputest.log
tail -5 gpurun_out/gputest.log
python tools/gemm_vs_blas.py > gpurun_out/blas_randn.txt 2>&1
ISO_DATA=zeros python tools/gemm_vs_blas.py > gpurun_out/blas_zeros.txt 2>&1
cat gpurun_out/blas_randn.txt gpurun_out/blas_zeros.txt
python bench.py > gpurun_out/bench_base.json 2> gpurun_out/bench_base.err; tail -c 3000 gpurun_out/bench_base.json

set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --depth 12 --batch 16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_d12.log 2>&1
tail -3 gpurun_out/b_d12.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d12 -o d12 -- python bench.py --depth 12 --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_d12.log 2>&1
ls gpurun_out/prof_d12/* | head

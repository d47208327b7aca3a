set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?" >> gpurun_out/bench1.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats -o r01 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_stats -name '*stats*' | head; find gpurun_out/prof_stats -type f | head -20
echo done

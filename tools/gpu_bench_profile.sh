# usage: bash tools/gpu_bench_profile.sh <tag>  -- bench line + rocprofv3 kernel stats + PMC passes for the headline workload
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/$TAG/stats.log 2>&1
cd $GRAFT_REPO_ROOT
bash tools/gpu_pmc.sh $TAG > gpurun_out/$TAG/pmc.log 2>&1
find gpurun_out/$TAG -name '*stats*.csv' | head

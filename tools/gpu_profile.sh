# usage: bash tools/gpu_profile.sh <tag>   (on the GPU box, from the repo root)
# bench line + rocprofv3 kernel stats + PMC passes (separate passes, kernel-trace only) for the headline workload in
# both stepping modes.  Nothing under rocprofv3 ever starts a child process (bench.py --no-build, tools/frames.py).
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py > $OUT/build.log 2>&1
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-build > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_b -o stats -- python3 $GRAFT_REPO_ROOT/tools/frames.py --mode 1 --frames 20 --hit-records 0 > $OUT/stats_b.log 2>&1
run() { dir=$1; name=$2; shift; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$dir/$name -o $name -- python3 $GRAFT_REPO_ROOT/tools/frames.py --mode $MODE --frames 3 --hit-records 0 > $OUT/$dir/$name.log 2>&1; }
for MODE in 0 1; do
  D=pmc_mode$MODE
  mkdir -p $OUT/$D
  run $D sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY
  run $D sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
  run $D fetch FETCH_SIZE
  run $D write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
  run $D grbm GRBM_GUI_ACTIVE GRBM_COUNT
  run $D tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc_mode0 raycast_svo_kernel > $OUT/pmc_mode0_summary.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_mode1 raycast_jump_kernel > $OUT/pmc_mode1_summary.txt 2>&1
tools/ubench/valu_issue > $OUT/valu_issue.txt 2>&1
echo done

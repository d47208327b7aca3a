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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-survey-camera --no-build > $OUT/stats.log 2>&1
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
  # VALU instructions by class (round 4: time-weighted issue estimate; the class of every opcode is calibrated on the ubench below)
  run $D vclass1 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
  run $D vclass2 SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64
  run $D vclass3 SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc_mode0 raycast_svo_kernel > $OUT/pmc_mode0_summary.txt 2>&1
python tools/pmc_summary.py $OUT/pmc_mode1 raycast_jump_kernel > $OUT/pmc_mode1_summary.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_issue tools/ubench/valu_issue.hip > $OUT/valu_issue_build.log 2>&1
tools/ubench/valu_issue > $OUT/valu_issue.txt 2>&1
# which class counter counts which opcode: the same counters over the ubench (one opcode per kernel)
mkdir -p $OUT/pmc_ubench; cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"; do
  tag=$(echo $set | cut -d' ' -f2); timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_ubench/$tag -o $tag -- $GRAFT_REPO_ROOT/tools/ubench/valu_issue > $OUT/pmc_ubench/$tag.log 2>&1; done
cd $GRAFT_REPO_ROOT; python tools/pmc_classify.py $OUT/pmc_ubench > $OUT/valu_classes.txt 2>&1
# the other instances: C4 (4K, 2 lights: multi-light instance), C2 (depth 10, primary only), headline without jumps
pmc2() { name=$1; shift; mkdir -p $OUT/pmc_$name; cd /tmp; for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | cut -d' ' -f1); timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$name/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/tools/frames.py --frames 3 --hit-records 0 "$@" > $OUT/pmc_$name/$tag.log 2>&1; done
  cd $GRAFT_REPO_ROOT; python tools/pmc_summary.py $OUT/pmc_$name raycast_svo_kernel > $OUT/pmc_${name}_summary.txt 2>&1; tail -1 $OUT/pmc_$name/FETCH_SIZE.log | cut -c1-160 >> $OUT/pmc_${name}_summary.txt; }
pmc2 c4_4k_2lights --width 3840 --height 2160 --lights 2
pmc2 c2_d10_primary --depth 10 --set shadow_rays=0
pmc2 headline_no_jumps --set jump_min_run=16777216
pmc2 headline_4lights --lights 4
echo done

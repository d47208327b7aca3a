# usage (GPU box, repo root): bash tools/gpu_pmc_quick.sh <tag> [frames.py args...]   -- the SQ counter set + traffic of the exact kernel, one pass each
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/tools/frames.py --frames 3 --hit-records 0 "$@" > $OUT/$tag.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT raycast_svo_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

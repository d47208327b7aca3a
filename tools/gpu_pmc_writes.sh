# usage (GPU box, repo root): bash tools/gpu_pmc_writes.sh <tag> [frames.py args...]
# Where the HBM writes of the exact kernel come from: write-side counters, one rocprofv3 --pmc pass each
# (kernel-trace only), for the product library and for gpurun_variants/libvrc_nostore.so (tools/build_variant.sh nostore
# -DVRC_NO_FRAME_STORE: the frame is computed and not stored), hit records off.
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for lib in product nostore; do
  if [ $lib = nostore ]; then export VRC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libvrc_nostore.so; [ -f $VRC_LIB_PATH ] || continue; else unset VRC_LIB_PATH; fi
  for set in "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_LDS" "TCC_WRITE_sum TCP_TCC_WRITE_REQ_sum TCC_NORMAL_WRITEBACK_sum" "FETCH_SIZE"; do
    tag=${lib}_$(echo $set | cut -d' ' -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o $tag -- python3 $GRAFT_REPO_ROOT/tools/frames.py --frames 3 --hit-records 0 "$@" > $OUT/$tag.log 2>&1
  done
done
unset VRC_LIB_PATH
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT raycast_svo_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

# usage (GPU box): bash tools/gpu_pmc_variants.sh <outdir> "<frames.py args>" name[:lib] ...
# VALU / SALU instruction counts and wave cycles of the SVO kernel per library variant (gpurun_variants/libvrc_<lib>.so, or the
# product library for lib "prod"), one rocprofv3 --pmc pass each; nothing under rocprofv3 starts a child process
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; ARGS=$2; shift; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for spec in "$@"; do
  name=${spec%%:*}; lib=${spec##*:}; extra=""
  case "$name" in *@*) extra="--set ${name##*@}"; ;; esac
  if [ "$lib" = "prod" ]; then unset VRC_LIB_PATH; else export VRC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_variants/libvrc_$lib.so; fi
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $OUT/$name/sq -o sq -- python3 $GRAFT_REPO_ROOT/tools/frames.py --frames 3 --hit-records 0 $ARGS $extra > $OUT/$name.log 2>&1
  echo "== $name"; tail -1 $OUT/$name.log | cut -c1-120
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$name raycast_svo_kernel
done

OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_l1
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() { name=$1; mode=$2; shift; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/m$mode/$name -o $name -- python3 $GRAFT_REPO_ROOT/tools/frames.py --mode $mode --frames 3 --hit-records 0 > $OUT/m${mode}_$name.log 2>&1; }
for MODE in 1 0; do
run tcp1 $MODE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 $MODE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum
run ta $MODE TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
run tcc2 $MODE TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum
run sq3 $MODE SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/m1 raycast_jump_kernel > $OUT/summary_mode1.txt 2>&1
python tools/pmc_summary.py $OUT/m0 raycast_svo_kernel > $OUT/summary_mode0.txt 2>&1
cat $OUT/summary_mode1.txt; tail -3 $OUT/m1_tcp1.log

# SQ counters of the loss kernels (one PMC pass per run); summary printed per kernel variant
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_loss
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-nms --loss-steps 2 --no-cpu-baseline > $OUT/set$i.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_loss_summary.py $OUT

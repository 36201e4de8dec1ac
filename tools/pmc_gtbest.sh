cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gtbest
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/tools/exp_gt_best_cost.py > $OUT/set$i.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT && python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/set*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'match_encode_kernel' in r['Kernel_Name'] and int(r['Grid_Size']) > 5000000:
            k = 'gtbest' if 'false, true' in r['Kernel_Name'] else 'plain'
            vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in vals:
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(vals[k].items())})
PY

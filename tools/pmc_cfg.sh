# SQ counters of one bench_configs.py configuration:  bash tools/pmc_cfg.sh <cfg> [kernel-substring]
cd /tmp && export TMPDIR=/tmp
CFG=${1:-4}
PAT=${2:-match_encode_kernel}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_cfg$CFG
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --cfg $CFG --iters 3 > $OUT/set$i.log 2>&1 || exit 1
done
cd $GRAFT_REPO_ROOT && python3 - "$OUT" "$PAT" <<'PY'
import collections, csv, glob, sys
vals = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/set*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
print({c: round(sum(v) / len(v)) for c, v in sorted(vals.items())})
PY

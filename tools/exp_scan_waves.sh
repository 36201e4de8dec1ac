# needs a library built with ODBOX_TUNING_HOOKS=1 (python -m object_detection_amd.build --force), see odbox_infer.hip tuning_env()
# score-scan time against resident waves per CU (LDS left for other kernels)
cd /tmp && export TMPDIR=/tmp
for w in 3 4 5 6; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/scanw$w
rm -rf $OUT; mkdir -p $OUT
ODBOX_SCAN_WAVES_PER_CU=$w timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --cfg 5 > $OUT/log.txt 2>&1 || exit 1
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$OUT/*/*kernel_stats.csv')[0])):
    if 'score_scan' in r['Name']: print('waves/CU $w scan avg us', float(r['AverageNs'])/1000)
"
done

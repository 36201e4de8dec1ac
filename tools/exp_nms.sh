# needs a library built with ODBOX_TUNING_HOOKS=1 (python -m object_detection_amd.build --force), see odbox_infer.hip tuning_env()
# NMS kernel time with one of its two paths switched off (ODBOX_NMS_DBG 4: no cooperative blocks, 8: no wave blocks)
cd /tmp && export TMPDIR=/tmp
for d in 0 4 8; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/nmsdbg$d
rm -rf $OUT; mkdir -p $OUT
ODBOX_NMS_DBG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --cfg 5 > $OUT/log.txt 2>&1 || exit 1
echo "dbg=$d $(grep -h 'nms_kernel' $OUT/*/*kernel_stats.csv | awk -F'",' '{print $2,$3,$4}' | tr -d '"' | awk '{print $NF}' | head -1)"
python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$OUT/*/*kernel_stats.csv')[0])):
    if 'nms_kernel' in r['Name']: print('   nms_kernel avg us', float(r['AverageNs'])/1000)
"
done

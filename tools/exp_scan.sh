# needs a library built with ODBOX_TUNING_HOOKS=1 (python -m object_detection_amd.build --force), see odbox_infer.hip tuning_env()
run() { echo "== $*"; env "$@" timeout -k 10 100 python bench.py --steps 5 --warmup 2 --nms-steps 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['decode_nms']['ms_per_step'])"; }
run A=1
run ODBOX_SCAN_DBG=1
run ODBOX_SCAN_LPR_LOG2=0
run ODBOX_SCAN_LPR_LOG2=2
run ODBOX_SCAN_LPR_LOG2=2 ODBOX_SCAN_STEPS=1
run ODBOX_SCAN_LPR_LOG2=3 ODBOX_SCAN_STEPS=1
run ODBOX_SCAN_LPR_LOG2=4

# needs object_detection_amd/build/libodbox_hooks.so (ODBOX_TUNING_HOOKS=1 build); prints the scan kernels' average us per variant
export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_hooks.so
run() { printf "%-60s" "$*"; env "$@" bash tools/prof_scan16.sh 2>/dev/null | grep score_scan | awk '{printf "%s %s   ", $3, $(NF-4)}' | sed 's/(anonymous//g'; echo; }
run A=1
run ODBOX_SCAN_LPR_LOG2=1 ODBOX_SCAN_STEPS=1
run ODBOX_SCAN_LPR_LOG2=2 ODBOX_SCAN_STEPS=1
run ODBOX_SCAN_LPR_LOG2=2 ODBOX_SCAN_STEPS=2
run ODBOX_SCAN_LPR_LOG2=1 ODBOX_SCAN_STEPS=1 ODBOX_SCAN_WAVES_PER_CU=8

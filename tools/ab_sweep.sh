# Same-box A/B of match_encode_sweep_kernel variants (tools/build_variant.sh with ODBOX_TUNING_HOOKS=1):
#   bash tools/ab_sweep.sh <cfg> <batch> <name>:<grid,grid,...> ...
cfg=$1; B=$2; shift 2
for spec in "$@"; do
  n=${spec%%:*}; g=${spec#*:}
  if [ "$n" = default ]; then unset ODBOX_LIBRARY; else export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_$n.so; fi
  echo "== $n"
  timeout -k 10 100 python3 tools/exp_sweep.py $cfg $B $g 2>/dev/null | grep "us per" || exit 1
done

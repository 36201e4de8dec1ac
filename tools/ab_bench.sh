# Same-box A/B of bench.py's headline (rotating outputs) for library variants:  bash tools/ab_bench.sh <name> ...
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset ODBOX_LIBRARY; else export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_$n.so; fi
    printf "%-18s" $n
    timeout -k 10 120 python3 bench.py --no-loss --no-nms --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read())
print('rotating %.2f us (median %.2f)   one set %.2f us (median %.2f)' % (r['roofline']['us_per_launch'], r['roofline']['us_per_launch_median'], r['roofline_cache_resident']['us_per_launch'], r['roofline_cache_resident']['us_per_launch_median']))"
  done
done

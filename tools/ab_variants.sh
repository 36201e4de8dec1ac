# Same-box A/B of library variants built with ODBOX_EXTRA_HIPCC_FLAGS (object_detection_amd/build/libodbox_<name>.so):
#   bash tools/ab_variants.sh "<cfg list>" <name> [<name> ...]      ("default" = the in-tree library)
CFGS=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    if [ "$n" = default ]; then unset ODBOX_LIBRARY; else export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_$n.so; fi
    printf "%-8s" $n
    timeout -k 10 120 python3 tools/bench_configs.py --cfg $CFGS 2>/dev/null | python3 -c "
import sys, json
print('  '.join('%s %.1f' % (str(r['cfg']).replace('2_loss_', ''), r['us']) for r in map(json.loads, sys.stdin) if 'autograd' not in str(r['cfg'])))"
  done
done

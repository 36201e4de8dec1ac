# forward loss step: one block per tile (base) / as many blocks as are resident, tiles dealt statically (static) / the
# same with the last round of tiles handed out by ticket (dyn, an experiment build); same box
for rep in 1 2 3; do
for v in base static dyn; do
  unset ODBOX_LIBRARY ODBOX_LOSS_DYN
  if [ $v = base ]; then export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_base.so; fi
  if [ $v = static ]; then export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_static.so; fi
  if [ $v = dyn ]; then export ODBOX_LIBRARY=$GRAFT_REPO_ROOT/object_detection_amd/build/libodbox_dyn.so ODBOX_LOSS_DYN=1; fi
  timeout -k 10 120 python bench.py --no-nms --no-configs --no-cpu-baseline --loss-steps 200 > gpurun_out/loss_x.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/loss_x.json').read().strip().splitlines()[-1]); ml=d['match_loss']
print('$v', ' '.join('%s %.2f' % (k, ml[k]['roofline']['us_per_launch_median']) for k in ml if isinstance(ml[k], dict) and 'roofline' in ml[k]))
PY
done; done

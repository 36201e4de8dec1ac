"""Same-process A/B of match_encode_sweep_kernel against the streaming tile kernel on cold (rotating) outputs.
Needs a library built with ODBOX_TUNING_HOOKS=1 (ODBOX_SWEEP / ODBOX_SWEEP_GRID are read per call).
    python tools/exp_sweep.py [cfg batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, anchors as A, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
c = workloads.CONFIGS[cfg]
anchors = bench.yolo_grid_anchors(416) if c['grid'] == 'yolo' else A.generate_anchors(A.AnchorsConfig(), c['S'])
gt_np, lab_np = workloads.match_inputs(cfg, batch=B, pad_to=int(os.environ['EXP_PAD_TO']) if os.environ.get('EXP_PAD_TO') else None)
gt, lab = torch.as_tensor(gt_np).to(dev), torch.as_tensor(lab_np).to(dev)
N, G, C, S = anchors.shape[0], gt.shape[1], c['C'], c['S']
per_set = B * N * (C + 6) * 4
sets = max(2, min(8, int(1.3e9 // per_set) + 1))
os.environ['ODBOX_SWEEP_MAX_MB'] = '4000'
regs = [torch.empty((B, N, 5), device=dev) for _ in range(sets)]
labs = [torch.empty((B, N, C + 1), device=dev) for _ in range(sets)]
print('cfg %d B %d N %d G %d: %.1f MB per launch, %d sets' % (cfg, B, N, G, per_set / 1e6, sets), flush=True)
lib.odbox_set_output_policy(2)
def step(i):
    k = i % sets
    assert lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                  (None if os.environ.get('EXP_NOREG') else regs[k].data_ptr()), labs[k].data_ptr(), None, None, None, None, st) == 0
grids = sys.argv[3].split(',') if len(sys.argv) > 3 else None
if grids:
    variants = [{'ODBOX_SWEEP': '0'}] + [{'ODBOX_SWEEP': '1', 'ODBOX_SWEEP_GRID': g} for g in grids]
else:
    variants = [{'ODBOX_SWEEP': '0'}, {'ODBOX_SWEEP': '1'}, {'ODBOX_SWEEP': '1', 'ODBOX_SWEEP_GT_LDS': '0'},
                {'ODBOX_SWEEP': '1', 'ODBOX_SWEEP_GRID': '256'}, {'ODBOX_SWEEP': '1', 'ODBOX_SWEEP_GRID': '384'}]
for rep in range(2):
    for v in variants:
        for k in ('ODBOX_SWEEP', 'ODBOX_SWEEP_GRID', 'ODBOX_SWEEP_GT_LDS'):
            os.environ.pop(k, None)
        os.environ.update(v)
        iters = 100 if per_set < 6e8 else 30
        _, ev = bench.timed(step, iters, 5, 1)
        print('%-50s %.2f us per launch' % (v, ev / iters * 1e6), flush=True)

"""Soak of the sweep's store / patch ordering: the same batch through match_encode_sweep_kernel many times into rotating,
dirty buffers (previous contents: ones, NaN, the last result), every result compared bit for bit with the tile kernel's.
    python tools/soak_sweep.py [iterations]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from object_detection_amd import _lib, anchors as A, workloads
lib = _lib.load(); dev = _lib.require_gpu(); st = _lib.stream_ptr()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for cfg, B, pad in ((2, 64, None), (1, 8, 64), (2, 150, 64), (2, 7, None)):
    c = workloads.CONFIGS[cfg]
    anchors = bench.yolo_grid_anchors(416) if c['grid'] == 'yolo' else A.generate_anchors(A.AnchorsConfig(), c['S'])
    gt_np, lab_np = workloads.match_inputs(cfg, batch=B, pad_to=pad)
    gt, lab = torch.as_tensor(gt_np).to(dev), torch.as_tensor(lab_np).to(dev)
    N, G, C, S = anchors.shape[0], gt.shape[1], c['C'], c['S']
    def call(reg, labs):
        assert lib.odbox_match_encode(anchors.data_ptr(), gt.data_ptr(), lab.data_ptr(), B, N, G, C, float(S), float(S), 0.4, 0.5,
                                      reg.data_ptr(), labs.data_ptr(), None, None, None, None, st) == 0
    ref_reg, ref_lab = torch.empty((B, N, 5), device=dev), torch.empty((B, N, C + 1), device=dev)
    lib.odbox_set_output_policy(1)
    call(ref_reg, ref_lab)
    assert lib.odbox_match_encode_last_plan() == 0
    lib.odbox_set_output_policy(2)
    sets = 3
    regs = [torch.ones((B, N, 5), device=dev) for _ in range(sets)]
    labs = [torch.full((B, N, C + 1), float('nan'), device=dev) for _ in range(sets)]
    ref_reg_i = ref_reg.view(torch.int32)
    for i in range(iters):
        k = i % sets
        if i % 7 == 3:
            labs[k].fill_(1.0)
        call(regs[k], labs[k])
        assert lib.odbox_match_encode_last_plan() == 2
        if not (torch.equal(labs[k], ref_lab) and torch.equal(regs[k].view(torch.int32), ref_reg_i)):
            bad += 1
            print('MISMATCH cfg %d B %d iteration %d: %d label elements differ' % (cfg, B, i, int((labs[k] != ref_lab).sum())), flush=True)
    lib.odbox_set_output_policy(0)
    print('cfg %d B %d G %d: %d iterations, %d mismatches so far' % (cfg, B, G, iters, bad), flush=True)
    del regs, labs, ref_reg, ref_lab
    torch.cuda.empty_cache()
sys.exit(1 if bad else 0)

"""One-off extension of tests/test_gpu_fuzz.py::test_fused_pass_one_block_random_shapes to seeds 32 .. 831 (800 more
batches through the fused pass-1 block of decode+NMS, the separate kernels and the C oracle), with a census of which
rung of the tier ladder finished each image (the block leaves it in the workspace: 0 = pass 2)."""
import collections
import importlib.util
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
spec = importlib.util.spec_from_file_location('fz', 'tests/test_gpu_fuzz.py')
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)
from object_detection_amd import _lib, bndbox as BB  # noqa: E402

census = collections.Counter()
plain_nms = BB.nms
fused_on = [True]
plain_switch = BB.set_nms_fused_tail


def switch(on):
    fused_on[0] = bool(on)
    return plain_switch(on)


def counting_nms(boxes, scores, **kw):
    out = plain_nms(boxes, scores, **kw)
    if fused_on[0]:
        B, N, C = scores.shape
        lib = _lib.load()
        ws = _lib.workspace('decode_nms', _lib.require_gpu(), lib.odbox_decode_nms_workspace(B, N, C, kw['max_ret'], 0))
        state = ws[256 + 2 * B * 64 * 4:][:4 * B].view(torch.int32).cpu().tolist()
        census.update('pass 2' if s == 0 else 'rung %d' % s for s in state)
    return out


BB.nms = counting_nms
BB.set_nms_fused_tail = switch
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 832)
bad = 0
for seed in range(lo, hi):
    try:
        fz.test_fused_pass_one_block_random_shapes(seed)
    except AssertionError as e:
        bad += 1
        print('seed', seed, 'FAILED', str(e)[:300], flush=True)
    if seed % 100 == 0:
        print('seed', seed, 'bad so far', bad, flush=True)
print('done, failures:', bad, ' images by rung:', dict(census))

import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from object_detection_amd import _lib, anchors as A, bndbox as BB, workloads
dev = _lib.require_gpu()
c = workloads.CONFIGS[5]
anchors = A.generate_anchors(A.AnchorsConfig(), c['S']); N = anchors.shape[0]
reg, sc = workloads.nms_inputs(N, C=c['C'], batch=64, p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=64)
reg, sc = torch.as_tensor(reg).to(dev), torch.as_tensor(sc).to(dev)
def call(): return BB.decode_and_nms(anchors, reg, sc, (c['S'], c['S']), pre_nms_topk=5000)
for _ in range(8): call()
torch.cuda.synchronize()
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(30):
        call(); ev[i + 1].record()
    torch.cuda.synchronize()
    print(' '.join('%.0f' % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(30)))

"""score-scan cache policy against tensor size: decode+NMS time with nt loads forced off / on
(needs a library built with ODBOX_TUNING_HOOKS=1, see odbox_infer.hip tuning_env())."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from object_detection_amd import anchors as A, bndbox as BB, workloads
    B = int(sys.argv[1])
    anchors = A.generate_anchors(A.AnchorsConfig(), 512)
    N = anchors.shape[0]
    reg, sc = workloads.nms_inputs(N, C=90, batch=B, distinct=min(B, 8), p3p4_anchors=9 * (64 * 64 + 32 * 32))
    reg, sc = torch.as_tensor(reg).cuda(), torch.as_tensor(sc).cuda()
    for _ in range(5):
        BB.decode_and_nms(anchors, reg, sc, (512, 512))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        BB.decode_and_nms(anchors, reg, sc, (512, 512))
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(B=B, MB=round(sc.numel() * 4 / 2**20), stream=os.environ.get('ODBOX_SCAN_STREAM'), us=round(e0.elapsed_time(e1) / 30 * 1000, 1))))
else:
    for B in (4, 8, 16, 32, 64):
        for s in ('0', '1'):
            subprocess.run([sys.executable, __file__, str(B)], env=dict(os.environ, ODBOX_SCAN_STREAM=s), check=True)

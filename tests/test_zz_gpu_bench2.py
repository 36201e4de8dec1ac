"""bench.py's N = 2 run, rehearsed on ONE GPU exactly as the driver launches it for N > 1
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py
--gpus 2 ...`), with ODBOX_DIST_BACKEND=gloo so that the two ranks -- both on cuda:0 -- exchange their records through
gloo instead of RCCL, which has no peer here.  What it protects: the launch contract, the process-group set-up, the
barrier / max-over-ranks timing, both ranks' HIP kernels and the gather of the detections, end to end and from a fresh
process; the numbers of such a run mean nothing.  Collected late on purpose: it starts child processes."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_ranks_on_one_gpu():
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env['ODBOX_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--nms-steps', '2', '--loss-steps', '2', '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                         # rank 0 only
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 128                      # 64 images per rank
    assert out['cpu_baseline'] is None                               # N = 1 only
    assert out['roofline']['kernel'] == out['roofline_auto_policy']['kernel'] == 'match_encode_sweep_kernel'
    nms = out['decode_nms']
    assert 'all-gather' in nms['metric'] and nms['config']['global_batch'] == 128
    assert nms['pipelined']['equals_single_calls'] is True
    assert nms['detections_per_image_mean'] > 10

    # the gathered detections: rank r decodes the batch of seed 1005 + 7919 r (bench.py), every rank ends up with
    # both shards in rank order -- the same bits as one process computing the two batches
    from object_detection_amd import anchors as A, bndbox as BB, workloads
    anchors = A.generate_anchors(A.AnchorsConfig(), 512)
    shards = []
    for rank in range(2):
        reg, sc = workloads.nms_inputs(anchors.shape[0], C=90, batch=64, seed=1005 + 7919 * rank,
                                       p3p4_anchors=9 * (64 * 64 + 32 * 32), distinct=64)
        shards.append(BB.decode_and_nms(anchors, reg, sc, (512, 512), pre_nms_topk=5000))
    digest = hashlib.sha1()
    for k in range(3):
        digest.update(torch.cat([s[k] for s in shards], 0).cpu().numpy().tobytes())
    assert nms['detections_sha1'] == digest.hexdigest()

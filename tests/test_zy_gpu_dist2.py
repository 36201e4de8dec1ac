"""The N > 1 path with the HIP kernels in it, rehearsed on ONE GPU: two ranks (gloo process group,
both on cuda:0) each decode + NMS their image shard and all-gather the records; every rank must end
up with exactly what a single process computes for the whole batch.  (The real multi-GPU run uses
RCCL; only the transport differs.)  Collected late on purpose: it starts child processes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
S, C = 256, 12


def _inputs(B):
    from object_detection_amd import workloads
    from oracle import odbox_oracle as O
    anchors = O.generate_anchors(S)
    reg, sc = workloads.nms_inputs(anchors.shape[0], C=C, batch=B, seed=31, clusters=10, distinct=B,
                                   p3p4_anchors=9 * (32 * 32 + 16 * 16))
    return anchors, reg, sc


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', ODBOX_DIST_BACKEND='gloo')       # the rehearsal switch: gloo group, host-staged records
    import torch.distributed as dist
    from object_detection_amd import dist as oddist
    try:
        oddist.init_from_env()
        anchors, reg, sc = _inputs(B)
        b, s, l = oddist.sharded_decode_and_nms(anchors, torch.as_tensor(reg).cuda(), torch.as_tensor(sc).cuda(), (S, S))
        q.put((rank, b.cpu().numpy(), s.cpu().numpy(), l.cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                              # pragma: no cover - reported by the parent
        q.put((rank, repr(e), None, None))


@pytest.mark.parametrize('B', [6, 5])                   # even shards, and a short last shard
def test_two_ranks_on_one_gpu_equal_one_process(B):
    from object_detection_amd import bndbox as BB
    anchors, reg, sc = _inputs(B)
    want = [t.cpu().numpy() for t in BB.decode_and_nms(anchors, reg, sc, (S, S))]
    assert (want[1] > 0).sum() > 10
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=150) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, b, s, l in res:
        assert not isinstance(b, str), b
        np.testing.assert_array_equal(b, want[0])
        np.testing.assert_array_equal(s, want[1])
        np.testing.assert_array_equal(l, want[2])


def _rccl_worker(port, B, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    import torch.distributed as dist
    from object_detection_amd import bndbox as BB, dist as oddist
    try:
        torch.cuda.set_device(0)
        dist.init_process_group(backend='nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        anchors, reg, sc = _inputs(B)
        b, s, l = BB.decode_and_nms(anchors, torch.as_tensor(reg).cuda(), torch.as_tensor(sc).cuda(), (S, S))
        # B - 1: the short-shard padding and the cut back to `batch` run too
        gb, gs, gl = oddist.gather_detections(b[:B - 1], s[:B - 1], l[:B - 1], batch=B - 1, always=True)
        torch.cuda.synchronize()
        q.put((dist.get_backend(), gb.cpu().numpy(), gs.cpu().numpy(), gl.cpu().numpy(),
               b.cpu().numpy(), s.cpu().numpy(), l.cpu().numpy()))
        dist.destroy_process_group()
    except Exception as e:                              # pragma: no cover - reported by the parent
        q.put((repr(e),) + (None,) * 6)


def test_rccl_transport_with_a_group_of_one():
    """The collective of the real multi-GPU run -- all_gather_into_tensor of device records over the "nccl"
    backend (RCCL) -- executed on this one GPU with a group of one rank: records packed by the HIP kernel, through
    RCCL, unpacked by the HIP kernel, equal to the detections that went in, bit for bit."""
    B = 5
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, B, q))
    p.start()
    backend, gb, gs, gl, b, s, l = q.get(timeout=150)
    p.join(60)
    assert gb is not None, backend
    assert backend == 'nccl'
    assert p.exitcode == 0
    assert (s > 0).sum() > 10
    np.testing.assert_array_equal(gb, b[:B - 1])
    np.testing.assert_array_equal(gs, s[:B - 1])
    np.testing.assert_array_equal(gl, l[:B - 1])

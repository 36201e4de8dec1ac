"""N > 1 path on the CPU: world_size-2 gloo process group, batch sharding and the single
all-gather of detection records (object_detection_amd.dist)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_detection_amd import dist as oddist


def test_shard_range_covers_batch():
    for B in (1, 7, 64, 512):
        for W in (1, 2, 4, 8):
            got = []
            for r in range(W):
                lo, hi = oddist.shard_range(B, r, W)
                assert 0 <= lo <= hi <= B
                got += list(range(lo, hi))
            assert got == list(range(B))


def test_pack_unpack_is_bit_preserving():
    b = torch.rand(3, 5, 4)
    s = torch.rand(3, 5)
    l = torch.randint(0, 90, (3, 5), dtype=torch.int32)
    rb, rs, rl = oddist.unpack_detections(oddist.pack_detections(b, s, l))
    assert torch.equal(rb, b) and torch.equal(rs, s) and torch.equal(rl, l) and rl.dtype == torch.int32


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, R, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w = oddist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    gen = torch.Generator().manual_seed(5)
    boxes = torch.rand(B, R, 4, generator=gen)
    scores = torch.rand(B, R, generator=gen)
    labels = torch.randint(0, 90, (B, R), generator=gen, dtype=torch.int32)
    lo, hi = oddist.shard_range(B, rank, world)
    gb, gs, gl = oddist.gather_detections(boxes[lo:hi], scores[lo:hi], labels[lo:hi], batch=B)
    ok = torch.equal(gb, boxes) and torch.equal(gs, scores) and torch.equal(gl, labels)
    q.put((rank, bool(ok), tuple(gb.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_detections_world2_gloo():
    ctx = mp.get_context('spawn')
    for B in (8, 7):                       # even split and a short last shard
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, B, 10, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=120) for _ in procs)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert res == [(0, True, (B, 10, 4)), (1, True, (B, 10, 4))]

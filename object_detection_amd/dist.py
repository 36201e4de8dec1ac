"""Multi-GPU plumbing: one process per GPU, images sharded by batch index, ONE all-gather of
fixed-size detection records per eval batch (RCCL over xGMI when the backend is "nccl").

The reference has no counterpart: under Horovod every rank evaluates the whole eval set on
its own (detection.py:333 shards only the training files) and runs COCOeval locally
(evaluate.py:164-190).  Match+encode needs no collective -- targets feed the local loss shard.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun) and pin
    this rank to GPU LOCAL_RANK.  Returns (rank, world_size).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            # ODBOX_DIST_BACKEND=gloo: rehearse the multi-rank path where RCCL has no peers (several ranks on one GPU)
            backend = os.environ.get('ODBOX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kwargs = {}
        if backend == 'nccl':
            kwargs['device_id'] = torch.device('cuda', torch.cuda.current_device())
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def shard_range(batch, rank, world):
    """Contiguous image range [lo, hi) of `rank`; the batch is cut into `world` equal slices of
    ceil(batch / world) images, trailing slices may be short or empty."""
    per = (batch + world - 1) // world
    lo = min(rank * per, batch)
    return lo, min(lo + per, batch)


def pack_detections(boxes, scores, labels, rows=None):
    """[b,R,4] f32, [b,R] f32, [b,R] i32 -> [rows or b, R, 6] f32 records (label bit-cast, not
    converted; rows beyond b are zero).  One HIP launch on the GPU; torch ops for CPU tensors (the
    gloo tests)."""
    b, R = int(scores.shape[0]), int(scores.shape[1])
    rows = b if rows is None else max(rows, b)
    if boxes.is_cuda:
        from . import _lib
        lib = _lib.load()
        boxes, scores, labels = boxes.contiguous(), scores.contiguous(), labels.contiguous()
        rec = (torch.zeros if rows > b else torch.empty)((rows, R, 6), dtype=torch.float32, device=boxes.device)
        _lib.check(lib.odbox_pack_records(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(labels), b * R, _lib.ptr(rec),
                                          _lib.stream_ptr()), 'odbox_pack_records')
        return rec
    rec = torch.cat([boxes, scores.unsqueeze(-1), labels.view(torch.float32).unsqueeze(-1)], dim=-1)
    if rows > b:
        rec = torch.cat([rec, torch.zeros((rows - b, R, 6), dtype=rec.dtype, device=rec.device)], dim=0)
    return rec.contiguous()


def unpack_detections(records):
    if records.is_cuda:
        from . import _lib
        lib = _lib.load()
        records = records.contiguous()
        n, R = int(records.shape[0]), int(records.shape[1])
        boxes = torch.empty((n, R, 4), dtype=torch.float32, device=records.device)
        scores = torch.empty((n, R), dtype=torch.float32, device=records.device)
        labels = torch.empty((n, R), dtype=torch.int32, device=records.device)
        _lib.check(lib.odbox_unpack_records(_lib.ptr(records), n * R, _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(labels),
                                            _lib.stream_ptr()), 'odbox_unpack_records')
        return boxes, scores, labels
    boxes = records[..., 0:4].contiguous()
    scores = records[..., 4].contiguous()
    labels = records[..., 5].contiguous().view(torch.int32)
    return boxes, scores, labels


def _rehearsal_gather_through_host(rec, world, group):
    """Device records on a gloo group: gloo moves host memory only.  This exists for ONE purpose, rehearsing the
    N > 1 path where RCCL has no peers (several ranks on one GPU: tests/test_zy_gpu_dist2.py, tests/test_zz_gpu_bench2.py),
    and only runs when the process says so with ODBOX_DIST_BACKEND=gloo -- the same switch that makes init_from_env
    open a gloo group.  Anywhere else it is an error: a real run must never fall into a blocking host copy."""
    if os.environ.get('ODBOX_DIST_BACKEND') != 'gloo':
        from . import _lib
        raise _lib.OdboxError('gather_detections: device tensors on a gloo process group; use the "nccl" backend (RCCL), '
                              'or set ODBOX_DIST_BACKEND=gloo for the host-staged rehearsal on a machine without peers')
    host = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype)
    dist.all_gather_into_tensor(host, rec.cpu(), group=group)
    return host.to(rec.device)


def gather_detections(boxes, scores, labels, batch=None, group=None, always=False):
    """All ranks pass the detections of their shard ([b,R,4], [b,R], [b,R] with b =
    ceil(batch / world), short shards zero-padded here) and get the whole batch back, images in
    rank order: one all_gather_into_tensor of [b,R,6] records.  A group of one rank returns its
    inputs untouched unless `always` is set (tests/test_zy_gpu_dist2.py runs the RCCL transport
    itself that way on a one-GPU machine)."""
    if not (dist.is_available() and dist.is_initialized()):
        return boxes, scores, labels
    if dist.get_world_size(group) == 1 and not always:
        return boxes, scores, labels
    world = dist.get_world_size(group)
    per = None if batch is None else (batch + world - 1) // world
    rec = pack_detections(boxes, scores, labels, rows=per)
    if rec.is_cuda and dist.get_backend(group) == 'gloo':
        out = _rehearsal_gather_through_host(rec, world, group)
    else:
        # the product path: device records, one collective (RCCL when the backend is "nccl"), no host copy
        out = torch.empty((world * rec.shape[0],) + tuple(rec.shape[1:]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(out, rec, group=group)
    if batch is not None:
        out = out[:batch]
    return unpack_detections(out)


def sharded_decode_and_nms(anchors, regressors, class_scores, im_size, **kw):
    """Every rank holds the full batch inputs (or at least its own slice filled in), computes
    decode + NMS for its slice only and all-gathers the records."""
    from . import bndbox
    if not (dist.is_available() and dist.is_initialized()):
        return bndbox.decode_and_nms(anchors, regressors, class_scores, im_size, **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    B = regressors.shape[0]
    lo, hi = shard_range(B, rank, world)
    b, s, l = bndbox.decode_and_nms(anchors, regressors[lo:hi], class_scores[lo:hi], im_size, **kw)
    return gather_detections(b, s, l, batch=B)

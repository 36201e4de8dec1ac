"""Drop-in for the reference module `bndbox` (bndbox.py) on MI355X.

Same names, argument order, defaults, return order and padding conventions as the reference;
torch tensors on the current ROCm device instead of tf.Tensor.  HIP kernels only (C-ABI of
include/odbox.h), no CPU fallback.
"""
import threading
from typing import Tuple

import torch

from . import _lib

def _boxes2(boxes, dev):
    boxes = _lib.as_dev(boxes, torch.float32, dev)
    if boxes.dim() != 2 or boxes.shape[1] != 4:
        raise ValueError('boxes must be [n,4]')
    return boxes


def scale_boxes(boxes, from_size: Tuple[int, int], to_size: Tuple[int, int]) -> torch.Tensor:
    """bndbox.py:10-41: x * (from_w / to_w), y * (from_h / to_h); sizes are (height, width)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _boxes2(boxes, dev)
    out = torch.empty_like(boxes)
    _lib.check(lib.odbox_scale_boxes(_lib.ptr(boxes), boxes.shape[0],
                                     from_size[1] / to_size[1], from_size[0] / to_size[0], 0,
                                     _lib.ptr(out), _lib.stream_ptr()), 'odbox_scale_boxes')
    return out


def normalize_bndboxes(boxes, image_size: Tuple[int, int]) -> torch.Tensor:
    """bndbox.py:44-68: x / (w - 1), y / (h - 1); image_size is (height, width)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _boxes2(boxes, dev)
    h, w = image_size
    out = torch.empty_like(boxes)
    _lib.check(lib.odbox_scale_boxes(_lib.ptr(boxes), boxes.shape[0], float(w - 1), float(h - 1), 1,
                                     _lib.ptr(out), _lib.stream_ptr()), 'odbox_scale_boxes')
    return out


def regress_bndboxes(boxes, regressors) -> torch.Tensor:
    """bndbox.py:71-99.  boxes [B,N,4] (the reference's tiled anchors) or [N,4] (shared by all
    images, no tile needed); regressors [B,N,4] -> [B,N,4]."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _lib.as_dev(boxes, torch.float32, dev)
    reg = _lib.as_dev(regressors, torch.float32, dev)
    if reg.dim() != 3 or reg.shape[2] != 4:
        raise ValueError('regressors must be [B,N,4]')
    B, N = int(reg.shape[0]), int(reg.shape[1])
    batched = boxes.dim() == 3
    if boxes.shape[-1] != 4 or boxes.shape[-2] != N or (batched and boxes.shape[0] != B):
        raise ValueError('boxes %r do not match regressors %r' % (tuple(boxes.shape), tuple(reg.shape)))
    out = torch.empty_like(reg)
    _lib.check(lib.odbox_regress_boxes(_lib.ptr(boxes), int(batched), _lib.ptr(reg), B, N,
                                       _lib.ptr(out), _lib.stream_ptr()), 'odbox_regress_boxes')
    return out


def clip_boxes(boxes, im_size: Tuple[int, int]) -> torch.Tensor:
    """bndbox.py:102-115: x in [0, w-1], y in [0, h-1]; im_size is (height, width)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _lib.as_dev(boxes, torch.float32, dev)
    if boxes.shape[-1] != 4:
        raise ValueError('boxes must be [...,4]')
    h, w = im_size
    out = torch.empty_like(boxes)
    _lib.check(lib.odbox_clip_boxes(_lib.ptr(boxes), boxes.numel() // 4, float(h), float(w),
                                    _lib.ptr(out), _lib.stream_ptr()), 'odbox_clip_boxes')
    return out


def bbox_overlap(bboxes, gt_bboxes) -> torch.Tensor:
    """bndbox.py:254-299: bboxes [B,N,4] (or [N,4] shared), gt_bboxes [B,G,4] -> IoU [B,N,G],
    -1 for padded (all-negative) GT rows."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    bb = _lib.as_dev(bboxes, torch.float32, dev)
    gt = _lib.as_dev(gt_bboxes, torch.float32, dev)
    if gt.dim() != 3 or gt.shape[2] != 4:
        raise ValueError('gt_bboxes must be [B,G,4]')
    B, G = int(gt.shape[0]), int(gt.shape[1])
    batched = bb.dim() == 3
    N = int(bb.shape[-2])
    if bb.shape[-1] != 4 or (batched and bb.shape[0] != B):
        raise ValueError('bboxes %r do not match gt_bboxes %r' % (tuple(bb.shape), tuple(gt.shape)))
    out = torch.empty((B, N, G), dtype=torch.float32, device=dev)
    _lib.check(lib.odbox_bbox_overlap(_lib.ptr(bb), _lib.ptr(gt), B, N, G, int(batched),
                                      _lib.ptr(out), _lib.stream_ptr()), 'odbox_bbox_overlap')
    return out


_KIND_DTYPE = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}       # ODBOX_DTYPE_* of include/odbox.h


def _half_kind(x):
    if isinstance(x, torch.Tensor):
        return 1 if x.dtype == torch.float16 else 2 if x.dtype == torch.bfloat16 else 0
    return 0


def _workspace(lib, dev, B, N, C, max_ret, topk):
    need = lib.odbox_decode_nms_workspace(B, N, C, max_ret, topk)
    if need == 0:
        raise _lib.OdboxError('odbox_decode_nms_workspace: unsupported sizes B=%d N=%d C=%d max_ret=%d'
                              % (B, N, C, max_ret))
    return _lib.workspace('decode_nms', dev, need)


def _decode_nms(anchors, regress_or_boxes, class_scores, im_size, do_clip,
                score_threshold, iou_threshold, max_ret, pre_nms_topk, extras):
    lib = _lib.load()
    dev = _lib.require_gpu()
    # 16-bit model outputs (autocast: bfloat16; the reference's mixed_float16: float16) go to the library as
    # they are and are widened where they are read (odbox_decode_nms_mixed): the result is the fp32 path's for
    # the widened tensors, at half the bytes of the score stream.  An odd class count has no 16-bit path
    # (two classes per dword): widened here, like everything that is not a 16-bit torch tensor.
    sc_dt = _half_kind(class_scores)
    if sc_dt and int(class_scores.shape[-1]) % 2:
        sc_dt = 0
    rb_dt = _half_kind(regress_or_boxes)
    rb = _lib.as_dev(regress_or_boxes, _KIND_DTYPE[rb_dt], dev)
    sc = _lib.as_dev(class_scores, _KIND_DTYPE[sc_dt], dev)
    if sc.dim() != 3 or rb.dim() != 3 or rb.shape[2] != 4 or rb.shape[:2] != sc.shape[:2]:
        raise ValueError('need boxes/regressors [B,N,4] and class_scores [B,N,C]; got %r, %r'
                         % (tuple(rb.shape), tuple(sc.shape)))
    B, N, C = (int(v) for v in sc.shape)
    if anchors is not None:
        anchors = _lib.as_dev(anchors, torch.float32, dev)
        if tuple(anchors.shape) != (N, 4):
            raise ValueError('anchors must be [N,4]')
    max_ret = int(max_ret)
    boxes = torch.empty((B, max_ret, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, max_ret), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_ret), dtype=torch.int32, device=dev)
    keep = torch.empty((B, max_ret), dtype=torch.int32, device=dev) if extras else None
    count = torch.empty((B,), dtype=torch.int32, device=dev) if extras else None
    if B > 0:
        ws = _workspace(lib, dev, B, N, C, max_ret, int(pre_nms_topk))
        h, w = (float(im_size[0]), float(im_size[1])) if im_size is not None else (0.0, 0.0)
        tail = (B, N, C, h, w, int(bool(do_clip)), float(score_threshold), float(iou_threshold), max_ret,
                int(pre_nms_topk), _lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(labels), _lib.ptr(keep), _lib.ptr(count),
                _lib.ptr(ws), ws.numel(), _lib.stream_ptr())
        if sc_dt or rb_dt:
            _lib.check(lib.odbox_decode_nms_mixed(_lib.ptr(anchors), _lib.ptr(rb), rb_dt, _lib.ptr(sc), sc_dt, *tail),
                       'odbox_decode_nms_mixed')
        else:
            _lib.check(lib.odbox_decode_nms(_lib.ptr(anchors), _lib.ptr(rb), _lib.ptr(sc), *tail), 'odbox_decode_nms')
    if extras:
        return boxes, scores, labels, keep, count
    return boxes, scores, labels


def set_nms_tier(factor: int) -> int:
    """The score tier of `nms` / `decode_and_nms` (`odbox_set_nms_tier`): pass 1 runs the NMS on every image's best
    factor * max_ret candidates only and pass 2 redoes the images whose tier kept fewer than max_ret boxes; default 8,
    0 = off.  Same detections either way.  Returns the previous factor."""
    lib = _lib.load()
    prev = lib.odbox_set_nms_tier(int(factor))
    if prev < 0:
        raise _lib.OdboxError(lib.odbox_last_error().decode())
    return prev


def set_nms_fused_tail(on: bool) -> bool:
    """Pass 1 of `nms` / `decode_and_nms` as one launch (`odbox_set_nms_fused_tail`; default on) or as the separate
    partition / NMS / final kernels.  Same detections either way.  Returns the previous setting."""
    lib = _lib.load()
    prev = lib.odbox_set_nms_fused_tail(1 if on else 0)
    if prev < 0:
        raise _lib.OdboxError(lib.odbox_last_error().decode())
    return bool(prev)


def nms(bboxes, class_scores, score_threshold: float = 0.3, iou_threshold: float = 0.45,
        max_ret: int = 100, pre_nms_topk: int = 0, return_keep: bool = False):
    """bndbox.py:117-252.  bboxes [B,N,4] xyxy, class_scores [B,N,C] ->
    (boxes [B,max_ret,4] xyxy, scores [B,max_ret], labels [B,max_ret] int32), score-sorted,
    zero-padded (pad score 0, pad label 0).  Each box gets ONE class (argmax), per-class greedy
    NMS keeps at most max_ret per class, then the global top max_ret.
    return_keep=True also returns (keep_anchor [B,max_ret] i32 with -1 padding, count [B])."""
    return _decode_nms(None, bboxes, class_scores, None, False, score_threshold, iou_threshold,
                       max_ret, pre_nms_topk, return_keep)


def decode_and_nms(anchors, regressors, class_scores, im_size, score_threshold: float = 0.3,
                   iou_threshold: float = 0.45, max_ret: int = 100, pre_nms_topk: int = 0,
                   return_keep: bool = False):
    """The inference tail of EfnDet.call (encoder.py:200-214) in one call: anchors [N,4] are
    broadcast over the batch (no tile), regress -> clip to im_size=(h, w) -> nms."""
    return _decode_nms(anchors, regressors, class_scores, im_size, True, score_threshold,
                       iou_threshold, max_ret, pre_nms_topk, return_keep)


class DecodeNmsPipeline:
    """Batches of the inference tail in flight on `depth` streams of the library's own (no counterpart in
    the reference, which decodes one batch at a time inside the model call, encoder.py:200-214).

    One call of `decode_and_nms` is a 178 us score scan (HBM-bound) followed by 55 us of partition / NMS /
    final top-k (latency-bound, memory system idle).  With the next batch on another stream the scan of
    batch k+1 runs beside the tail of batch k -- the scan kernel leaves LDS on every CU for exactly that and
    the tail kernels fit into it -- and the step time drops to about the scan's (231 -> 182 us per 64 images
    of the configs[4] share, `tools/exp_two_streams.py`).  Results are bit-identical to single calls.

        pipe = DecodeNmsPipeline()
        pending = [pipe.submit(anchors, reg, scores, (h, w)) for reg, scores in batches]
        for p in pending: boxes, scores, labels = p.result()      # ordered on the caller's current stream

    `submit` orders the side stream behind the caller's current stream (the inputs' producer) and returns at
    once; `result()` makes the caller's current stream wait for that batch (no host synchronisation)."""

    _pool = {}
    _pool_lock = threading.Lock()

    class _Pending:
        def __init__(self, outs, event):
            self._outs, self._event = outs, event

        def result(self):
            torch.cuda.current_stream().wait_event(self._event)
            for t in self._outs:
                t.record_stream(torch.cuda.current_stream())
            return self._outs

    def __init__(self, depth: int = 2, device=None):
        if depth < 1:
            raise ValueError('depth must be >= 1')
        dev = _lib.require_gpu() if device is None else torch.device(device)
        # side streams come from one pool per device, shared by every pipeline of the process: a process has few
        # hardware queues (4 by default, GPU_MAX_HW_QUEUES), streams beyond them share a queue and what they
        # carry runs one after the other -- two pipelines with streams of their own (2 + 3, plus the default
        # stream) measured 141 us per bf16 batch where one pipeline alone ran 108
        with DecodeNmsPipeline._pool_lock:
            pool = DecodeNmsPipeline._pool.setdefault((dev.type, dev.index), [])
            while len(pool) < depth:
                pool.append(torch.cuda.Stream(dev))
            self._streams = pool[:depth]
        self._next = 0

    def submit(self, anchors, regressors, class_scores, im_size, score_threshold: float = 0.3,
               iou_threshold: float = 0.45, max_ret: int = 100, pre_nms_topk: int = 0, return_keep: bool = False,
               inputs_ready: bool = False):
        """Same arguments as `decode_and_nms`.  inputs_ready=True skips the ordering behind the caller's
        stream (inputs that are known to be complete, e.g. resident weights of a benchmark)."""
        st = self._streams[self._next % len(self._streams)]
        self._next += 1
        if not inputs_ready:
            st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs = _decode_nms(anchors, regressors, class_scores, im_size, True, score_threshold, iou_threshold,
                               max_ret, pre_nms_topk, return_keep)
            ev = torch.cuda.Event()
            ev.record(st)
        for t in (anchors, regressors, class_scores):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        return DecodeNmsPipeline._Pending(outs, ev)

"""Drop-in for the reference module `anchors` (anchors.py) on MI355X.

Same names, argument order, defaults and return order as the reference; tensors are torch
tensors on the current ROCm device instead of tf.Tensor (NumPy arrays and CPU tensors are
accepted and uploaded).  All arithmetic runs in hand-written HIP kernels through the C-ABI of
include/odbox.h -- there is no CPU fallback, every call raises OdboxError without a GPU.
"""
import math
import typing

import numpy as np
import torch

from . import _lib
from .config import AnchorsConfig

_INF = float('inf')


class AnchorGenerator(object):
    """anchors.py:22-114.  `anchors` ([A,4] fp32, A = 3*len(aspect_ratios)) is computed on the
    host in float64 and rounded to fp32 last, exactly like the reference's NumPy code; tiling
    over a feature map is the HIP kernel `odbox_anchor_tile`."""

    def __init__(self, size: float, aspect_ratios: typing.Sequence[float], stride: int = 1):
        self.size = size
        self.stride = stride
        self.aspect_ratios = aspect_ratios
        self.anchor_scales = [2 ** 0, 2 ** (1 / 3.0), 2 ** (2 / 3.0)]
        self.anchors = self._generate()
        self._dev_cache = {}

    def __len__(self):
        return len(self.aspect_ratios) * len(self.anchor_scales)

    def __call__(self, *args, **kwargs):
        return self.tile_anchors_over_feature_map(*args, **kwargs)

    def _generate(self) -> torch.Tensor:
        # anchor a: ratio a // 3 (h / w), scale a % 3; equal-area boxes centred on the origin
        rows = []
        n_scales = len(self.anchor_scales)
        for a in range(len(self)):
            ratio = float(self.aspect_ratios[a // n_scales])
            side = self.size * self.anchor_scales[a % n_scales]
            w = math.sqrt((side * side) / ratio)
            h = w * ratio
            rows.append((0.0 - w * 0.5, 0.0 - h * 0.5, w - w * 0.5, h - h * 0.5))
        return torch.from_numpy(np.asarray(rows, dtype=np.float64).astype(np.float32))

    def tile_anchors_over_feature_map(self, feature_map_shape) -> torch.Tensor:
        """[H*W*A, 4] fp32 xyxy; cell (i, j) -> rows (i*W + j)*A .. +A (anchors.py:47-87)."""
        lib = _lib.load()
        dev = _lib.require_gpu()
        fh, fw = int(feature_map_shape[0]), int(feature_map_shape[1])
        base = self._dev_cache.get(dev)
        if base is None:
            base = self._dev_cache[dev] = _lib.as_dev(self.anchors, torch.float32, dev)
        out = torch.empty((fh * fw * len(self), 4), dtype=torch.float32, device=dev)
        _lib.check(lib.odbox_anchor_tile(_lib.ptr(base), len(self), fh, fw, float(self.stride),
                                         _lib.ptr(out), _lib.stream_ptr()), 'odbox_anchor_tile')
        return out


def generate_anchors(anchors_config: AnchorsConfig, im_shape: int) -> torch.Tensor:
    """detection.py:223-236: P3..P7 generators, feature side im_shape // 2**level, concat."""
    parts = []
    for level in range(3, 8):
        gen = AnchorGenerator(size=anchors_config.sizes[level - 3],
                              aspect_ratios=anchors_config.ratios,
                              stride=anchors_config.strides[level - 3])
        side = im_shape // (2 ** level)
        parts.append(gen((side, side, 3)))
    return torch.cat(parts, dim=0)


def _image_hw(images):
    """The reference reads only tf.shape(images)[1:3] (anchors.py:164-167).  Accept a tensor /
    array of shape [B,H,W,C], or the shape itself, or (H, W)."""
    shape = tuple(images.shape) if hasattr(images, 'shape') else tuple(images)
    if len(shape) == 2:
        return float(shape[0]), float(shape[1])
    if len(shape) == 4:
        return float(shape[1]), float(shape[2])
    raise ValueError('images must be [B,H,W,C] (or its shape, or (H, W)); got %r' % (shape,))


_POLICIES = {'auto': 0, 'cache': 1, 'stream': 2}


def set_output_policy(policy: str) -> str:
    """How `anchor_targets_bbox` / `match_encode` write their dense outputs (`odbox_set_output_policy`):
    'auto' (the default; a function of the call's shape alone, which assumes a buffer that is not in the 256 MB
    Infinity Cache: label tensors beyond 512 MB and the label-stream shapes the lockstep sweep serves -- wide rows,
    at most 64 GT rows -- go around the caches, everything else through them; `match_encode_plan` tells which kernel a
    shape gets, and a captured HIP graph holds exactly that one), 'cache' (the consumer reads the targets straight
    away, or one buffer is rewritten in a tight loop), or 'stream' (targets nothing reads soon).  Same bits either
    way.  Returns the previous policy."""
    if policy not in _POLICIES:
        raise ValueError("policy must be 'auto', 'cache' or 'stream'")
    prev = _lib.load().odbox_set_output_policy(_POLICIES[policy])
    return {v: k for k, v in _POLICIES.items()}[prev]


_PLANS = {0: 'tiles', 1: 'tiles_stream', 2: 'sweep'}


def match_encode_plan(batch: int, num_anchors: int, num_gt: int, num_classes: int,
                      labels_aligned16: bool = True, gt_best: bool = False) -> str:
    """Which kernel `match_encode` launches for dense labels of this shape under the current output
    policy (`odbox_match_encode_plan`): 'tiles', 'tiles_stream' or 'sweep'.  Same bits on every plan."""
    lib = _lib.load()
    rc = lib.odbox_match_encode_plan(batch, num_anchors, num_gt, num_classes, int(labels_aligned16), int(gt_best))
    if rc < 0:
        raise _lib.OdboxError(lib.odbox_last_error().decode())
    return _PLANS[rc]


def match_encode_last_plan():
    """The kernel this thread's last `match_encode` / `anchor_targets_bbox` call really launched, or None."""
    return _PLANS.get(_lib.load().odbox_match_encode_last_plan())


def match_encode(anchors, image_hw, bndboxes, labels, num_classes,
                 negative_overlap=0.4, positive_overlap=0.5,
                 dense=True, compact=False, gt_best=False):
    """Fused matcher + encoder (`odbox_match_encode`).  Returns a dict with `regression`
    [B,N,5] / `labels` [B,N,C+1] when dense, `match_idx` i32 / `state` i8 / `max_iou` f32
    [B,N] when compact, and `gt_best_anchor` i32 / `gt_best_iou` f32 [B,G] (views of one record
    tensor) when gt_best: per ground-truth row the anchor with the largest IoU, ties to the lowest
    anchor -- an extra output that never changes the others."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    anchors = _lib.as_dev(anchors, torch.float32, dev)
    bndboxes = _lib.as_dev(bndboxes, torch.float32, dev)      # fp16 GT is cast (anchors.py:251)
    if anchors.dim() != 2 or anchors.shape[1] != 4:
        raise ValueError('anchors must be [N,4]')
    if bndboxes.dim() != 3 or bndboxes.shape[2] != 4:
        raise ValueError('bndboxes must be [B,G,4]')
    B, G = int(bndboxes.shape[0]), int(bndboxes.shape[1])
    N = int(anchors.shape[0])
    C = int(num_classes)
    if labels is not None:
        labels = _lib.as_dev(labels, torch.int32, dev)
        if tuple(labels.shape) != (B, G):
            raise ValueError('labels must be [B,G]')
    out = {}
    reg = lab = midx = state = miou = None
    if dense:
        reg = out['regression'] = torch.empty((B, N, 5), dtype=torch.float32, device=dev)
        lab = out['labels'] = torch.empty((B, N, C + 1), dtype=torch.float32, device=dev)
    if compact:
        midx = out['match_idx'] = torch.empty((B, N), dtype=torch.int32, device=dev)
        state = out['state'] = torch.empty((B, N), dtype=torch.int8, device=dev)
        miou = out['max_iou'] = torch.empty((B, N), dtype=torch.float32, device=dev)
    best = None
    if gt_best:
        best = torch.empty((B, G, 2), dtype=torch.int32, device=dev)          # odbox_gt_best_t records
        out['gt_best_anchor'] = best[..., 0]
        out['gt_best_iou'] = best.view(torch.float32)[..., 1]
    _lib.check(lib.odbox_match_encode(
        _lib.ptr(anchors), _lib.ptr(bndboxes), _lib.ptr(labels), B, N, G, C,
        float(image_hw[0]), float(image_hw[1]), float(negative_overlap), float(positive_overlap),
        _lib.ptr(reg), _lib.ptr(lab), _lib.ptr(midx), _lib.ptr(state), _lib.ptr(miou), _lib.ptr(best),
        _lib.stream_ptr()), 'odbox_match_encode')
    return out


def gt_best_anchors(anchors, annotations):
    """Per ground-truth row, the anchor with the largest IoU and that IoU: `(best_anchor [B,G] int32,
    best_iou [B,G] float32)` = argmax / max of `bbox_overlap(anchors, annotations)` over its anchor
    axis (ties to the lowest anchor; padded rows give (0, -1)).  The reference has no forced
    per-GT match (anchors.py:253-272); its only per-GT argmax is the diagnostic of coco.py:399-402."""
    r = match_encode(anchors, (_INF, _INF), annotations, None, 1, dense=False, compact=False, gt_best=True)
    return r['gt_best_anchor'], r['gt_best_iou']


def anchor_targets_bbox(anchors, images, bndboxes, labels, num_classes: int,
                        negative_overlap: float = 0.4, positive_overlap: float = 0.5,
                        dtype=torch.float32) -> typing.Tuple[torch.Tensor, torch.Tensor]:
    """anchors.py:116-213.  Returns (regression [B,N,5], labels [B,N,num_classes+1]) -- in that
    order, like the reference's return statement (:213; its docstring has them swapped).
    Last column of both: anchor state -1 ignore / 0 background / 1 foreground.
    `dtype` only casts the image height/width in the reference; fp32 here as there."""
    r = match_encode(anchors, _image_hw(images), bndboxes, labels, num_classes,
                     negative_overlap, positive_overlap, dense=True, compact=False)
    return r['regression'], r['labels']


def compute_gt_annotations(anchors, annotations, negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:215-274 -> (positive [B,N] bool, ignore [B,N] bool, batched_indices [B*N,2] i32).
    Pure per-anchor argmax + two thresholds, no out-of-image test, no per-GT forced match."""
    r = match_encode(anchors, (_INF, _INF), annotations, None, 1,
                     negative_overlap, positive_overlap, dense=False, compact=True)
    state = r['state']
    B, N = state.shape
    positive = state == 1
    ignore = state == -1
    bidx = torch.arange(B, dtype=torch.int32, device=state.device).repeat_interleave(N)
    batched = torch.stack([bidx, r['match_idx'].reshape(-1)], dim=1)
    return positive, ignore, batched


def bbox_transform(anchors, gt_boxes) -> torch.Tensor:
    """anchors.py:277-299: anchors [N,4] or [B,N,4], gt_boxes [B,N,4] (or [N,4]) -> offsets."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    anchors = _lib.as_dev(anchors, torch.float32, dev)
    gt = _lib.as_dev(gt_boxes, torch.float32, dev)
    squeeze = gt.dim() == 2
    if squeeze:
        gt = gt.unsqueeze(0)
    B, N = int(gt.shape[0]), int(gt.shape[1])
    batched = anchors.dim() == 3
    if anchors.shape[-2] != N or (batched and anchors.shape[0] != B):
        raise ValueError('anchors %r do not broadcast against gt_boxes %r'
                         % (tuple(anchors.shape), tuple(gt.shape)))
    out = torch.empty_like(gt)
    _lib.check(lib.odbox_bbox_transform(_lib.ptr(anchors), int(batched), _lib.ptr(gt), B, N,
                                        _lib.ptr(out), _lib.stream_ptr()), 'odbox_bbox_transform')
    return out[0] if squeeze else out

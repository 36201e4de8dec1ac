"""Drop-in for the ground-truth side of the reference module `detection` (detection.py) on MI355X
(SURVEY.md §8f row 3): the box arithmetic of `prepare_example` (:97-141), `polygon2bbox` (:77-95)
and the `padded_batch` of `wrap_dataset` (:257-262) as ONE kernel launch per batch, from ragged
per-image annotation lists to the dense `[B,max_items,4]` / `[B,max_items]` tensors the matcher
reads.  Image decoding, augmentation and tf.data plumbing are out of scope.  HIP only, no CPU
fallback.  `generate_anchors` (:223-236) lives in `anchors.py` and is re-exported here."""
import numpy as np
import torch

from . import _lib
from .anchors import generate_anchors  # noqa: F401

MAX_ITEMS_IN_IMAGE = 64          # detection.py:33 --max_items_in_image


def _ragged(rows, width, dev):
    """list of per-image arrays [n_i, *width] -> (flat device tensor, device offsets [B+1])."""
    counts = [int(np.shape(r)[0]) if np.size(r) else 0 for r in rows]
    offs = np.zeros(len(rows) + 1, dtype=np.int32)
    np.cumsum(counts, out=offs[1:])
    parts = [np.asarray(r, dtype=np.float32).reshape((-1,) + width) for r, c in zip(rows, counts) if c]
    flat = np.concatenate(parts) if parts else np.zeros((0,) + width, np.float32)
    return _lib.as_dev(flat, torch.float32, dev), _lib.as_dev(offs, torch.int32, dev), counts


def _labels(labels, counts, dev):
    parts = [np.asarray(l, dtype=np.int32).reshape(-1) for l, c in zip(labels, counts) if c]
    for l, c in zip(labels, counts):
        if np.size(l) != c:
            raise ValueError('every image needs one label per box')
    return _lib.as_dev(np.concatenate(parts) if parts else np.zeros(0, np.int32), torch.int32, dev)


def _launch(entry, src, lab, offs, sz, B, image_size, max_items, dtype, want_yx, want_wh):
    lib = _lib.load()
    dev = src.device
    out_b = torch.empty((B, max_items, 4), dtype=torch.float32, device=dev)
    out_l = torch.empty((B, max_items), dtype=torch.int32, device=dev)
    mode = 1 if want_yx else (2 if want_wh else 0)
    fn = getattr(lib, entry)
    _lib.check(fn(_lib.ptr(src), _lib.ptr(lab), _lib.ptr(offs), int(lab.shape[0]), _lib.ptr(sz), B, int(max_items), float(image_size),
                  1 if dtype == torch.float16 else 0, mode, _lib.ptr(out_b), _lib.ptr(out_l), _lib.stream_ptr()), entry)
    return (out_b.to(dtype) if dtype != torch.float32 else out_b), out_l


def _prepare(entry, rows, width, labels, sizes, image_size, max_items, dtype, want_yx, want_wh):
    _lib.load()
    dev = _lib.require_gpu()
    if dtype not in (torch.float32, torch.float16):
        raise ValueError('dtype must be torch.float32 or torch.float16')
    B = len(rows)
    if len(labels) != B or len(sizes) != B:
        raise ValueError('boxes, labels and sizes must have one entry per image')
    src, offs, counts = _ragged(rows, width, dev)
    lab = _labels(labels, counts, dev)
    sz = _lib.as_dev(np.asarray(sizes, dtype=np.float32), torch.float32, dev)
    return _launch(entry, src, lab, offs, sz, B, image_size, max_items, dtype, want_yx, want_wh)


def prepare_batch(orig_bboxes, orig_labels, orig_image_hw, image_size, max_items=MAX_ITEMS_IN_IMAGE,
                  dtype=torch.float32, want_yx=False, want_wh=False):
    """`prepare_example` (evaluation path) + `padded_batch` for a whole batch.

    orig_bboxes: list (one entry per image) of `[n_i,4]` arrays `(x0, y0, w, h)` in original pixels;
    orig_labels: list of `[n_i]` int labels; orig_image_hw: `[B,2]` (height, width) of the decoded
    images.  Returns `(true_bboxes [B,max_items,4], true_labels [B,max_items] int32)`, rows past an
    image's own count filled with -1 (detection.py:262)."""
    return _prepare('odbox_gt_prepare_xywh', orig_bboxes, (4,), orig_labels, orig_image_hw, image_size, max_items,
                    dtype, want_yx, want_wh)


def prepare_batch_polygons(word_polys, labels, canvas_sizes, image_size, max_items=MAX_ITEMS_IN_IMAGE,
                           dtype=torch.float32, want_yx=False, want_wh=False):
    """detection.py:131-141 + padded_batch from augmented corner polygons: word_polys is a list of
    `[n_i,4,2]` arrays on a square canvas of side canvas_sizes[b] (`tf.shape(image)[1]`, :131)."""
    return _prepare('odbox_gt_prepare_poly', word_polys, (4, 2), labels, canvas_sizes, image_size, max_items,
                    dtype, want_yx, want_wh)


def polygon2bbox(poly, want_yx=False, want_wh=False):
    """detection.py:77-95: poly `[N,4,2]` -> `[N,4]` in poly's dtype (fp16 stays fp16)."""
    _lib.load()
    dev = _lib.require_gpu()
    dtype = torch.float16 if isinstance(poly, torch.Tensor) and poly.dtype == torch.float16 else torch.float32
    p = _lib.as_dev(poly, torch.float32, dev).reshape(-1, 4, 2)
    n = p.shape[0]
    if n == 0:
        return torch.empty((0, 4), dtype=dtype, device=dev)
    offs = torch.tensor([0, n], dtype=torch.int32, device=dev)
    lab = torch.zeros(n, dtype=torch.int32, device=dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)            # canvas 1 -> size 1: coordinates unchanged
    boxes, _ = _launch('odbox_gt_prepare_poly', p, lab, offs, one, 1, 1.0, n, dtype, want_yx, want_wh)
    return boxes[0]


def restore_detections(coords_batch, scores_batch, orig_hw, image_size: int):
    """infdet.py:140-174 for a whole batch in one launch (`odbox_restore_boxes`): detections of the
    padded, resized square back to original image pixels, in float64 like the reference's Python
    floats.  coords_batch [B,R,4] / scores_batch [B,R] as `bndbox.nms` returns them, orig_hw [B,2]
    = (height, width) of each original image.  Returns (boxes [B,R,4] float64 on the device, zero
    from the first score <= 0 on, counts [B] int32 = where the reference's loop breaks)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    boxes = _lib.as_dev(coords_batch, torch.float32, dev)
    scores = _lib.as_dev(scores_batch, torch.float32, dev)
    hw = _lib.as_dev(orig_hw, torch.int32, dev)
    B, R = int(scores.shape[0]), int(scores.shape[1])
    if tuple(boxes.shape) != (B, R, 4) or tuple(hw.shape) != (B, 2):
        raise ValueError('shape mismatch: coords %r, scores %r, orig_hw %r' % (tuple(boxes.shape), tuple(scores.shape), tuple(hw.shape)))
    out = torch.empty((B, R, 4), dtype=torch.float64, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(lib.odbox_restore_boxes(_lib.ptr(boxes), _lib.ptr(scores), _lib.ptr(hw), B, R, int(image_size),
                                       _lib.ptr(out), _lib.ptr(counts), _lib.stream_ptr()), 'odbox_restore_boxes')
    return out, counts


def results_json(filenames, boxes64, scores_batch, cat_ids_batch, counts):
    """The list infdet.py:152-188 dumps into results.json, from `restore_detections`' outputs (one
    device-to-host copy per tensor instead of one `.numpy()` per detection)."""
    boxes64 = torch.as_tensor(boxes64).cpu().numpy()
    scores = torch.as_tensor(scores_batch).cpu().numpy()
    cats = torch.as_tensor(cat_ids_batch).cpu().numpy()
    counts = torch.as_tensor(counts).cpu().numpy()
    out = []
    for b, filename in enumerate(filenames):
        anns = [{'bbox': [float(v) for v in boxes64[b, r]], 'class_score': float(scores[b, r]),
                 'category_id': int(cats[b, r])} for r in range(int(counts[b]))]
        out.append({'filename': filename, 'annotations': anns})
    return out

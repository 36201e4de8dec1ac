"""ctypes front-end of oracle/_build/libodbox_oracle.so (built from odbox_oracle.c by
`make -C oracle`).  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libodbox_oracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int32)
_b = ctypes.POINTER(ctypes.c_int8)


def build():
    subprocess.run(['make', '-C', _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.odbox_oracle_max_threads.restype = ctypes.c_int
        _lib.odbox_oracle_decode_nms.restype = ctypes.c_int
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def max_threads():
    return int(lib().odbox_oracle_max_threads())


def anchor_tile(base, fh, fw, stride):
    base = _c(base, np.float32)
    A = base.shape[0]
    out = np.empty((fh * fw * A, 4), dtype=np.float32)
    lib().odbox_oracle_anchor_tile(_p(base, _f), A, fh, fw, ctypes.c_float(stride), _p(out, _f))
    return out


def bbox_overlap(anchors, gt):
    anchors = _c(anchors, np.float32)
    gt = _c(gt, np.float32)
    batched = anchors.ndim == 3
    B, G = gt.shape[0], gt.shape[1]
    N = anchors.shape[-2]
    out = np.empty((B, N, G), dtype=np.float32)
    lib().odbox_oracle_bbox_overlap(_p(anchors, _f), _p(gt, _f), B, N, G, int(batched), _p(out, _f))
    return out


def match_encode(anchors, image_hw, gt_boxes, gt_labels, num_classes,
                 negative_overlap=0.4, positive_overlap=0.5, threads=1, dense=True):
    anchors = _c(anchors, np.float32)
    gt_boxes = _c(gt_boxes, np.float32)
    gt_labels = _c(gt_labels, np.int32)
    B, G = gt_labels.shape
    N = anchors.shape[0]
    C = int(num_classes)
    reg = np.empty((B, N, 5), dtype=np.float32) if dense else None
    lab = np.empty((B, N, C + 1), dtype=np.float32) if dense else None
    midx = np.empty((B, N), dtype=np.int32)
    state = np.empty((B, N), dtype=np.int8)
    miou = np.empty((B, N), dtype=np.float32)
    lib().odbox_oracle_match_encode(
        _p(anchors, _f), _p(gt_boxes, _f), _p(gt_labels, _i), B, N, G, C,
        ctypes.c_float(image_hw[0]), ctypes.c_float(image_hw[1]),
        ctypes.c_float(negative_overlap), ctypes.c_float(positive_overlap),
        _p(reg, _f), _p(lab, _f), _p(midx, _i), _p(state, _b), _p(miou, _f), int(threads))
    return {'regression': reg, 'labels': lab, 'match_idx': midx, 'state': state, 'max_iou': miou}


def gt_best(anchors, gt_boxes, threads=1):
    """(best_anchor [B,G] i32, best_iou [B,G] f32): column argmax / max of the IoU matrix."""
    anchors = _c(anchors, np.float32)
    gt_boxes = _c(gt_boxes, np.float32)
    B, G = gt_boxes.shape[0], gt_boxes.shape[1]
    ba = np.empty((B, G), dtype=np.int32)
    bi = np.empty((B, G), dtype=np.float32)
    lib().odbox_oracle_gt_best(_p(anchors, _f), _p(gt_boxes, _f), B, anchors.shape[0], G, _p(ba, _i), _p(bi, _f), int(threads))
    return ba, bi


def regress_boxes(boxes, regress):
    boxes = _c(boxes, np.float32)
    regress = _c(regress, np.float32)
    B, N = regress.shape[0], regress.shape[1]
    out = np.empty_like(regress)
    lib().odbox_oracle_regress_boxes(_p(boxes, _f), int(boxes.ndim == 3), _p(regress, _f), B, N,
                                     _p(out, _f))
    return out


def clip_boxes(boxes, image_hw):
    boxes = _c(boxes, np.float32)
    out = np.empty_like(boxes)
    lib().odbox_oracle_clip_boxes(_p(boxes, _f), ctypes.c_size_t(boxes.size // 4),
                                  ctypes.c_float(image_hw[0]), ctypes.c_float(image_hw[1]),
                                  _p(out, _f))
    return out


def decode_nms(anchors, regress_or_boxes, scores, image_hw, do_clip=True,
               score_threshold=0.3, iou_threshold=0.45, max_ret=100, threads=1):
    anchors = None if anchors is None else _c(anchors, np.float32)
    rb = _c(regress_or_boxes, np.float32)
    scores = _c(scores, np.float32)
    B, N, C = scores.shape
    ob = np.empty((B, max_ret, 4), dtype=np.float32)
    os_ = np.empty((B, max_ret), dtype=np.float32)
    ol = np.empty((B, max_ret), dtype=np.int32)
    ok = np.empty((B, max_ret), dtype=np.int32)
    on = np.empty((B,), dtype=np.int32)
    rc = lib().odbox_oracle_decode_nms(
        _p(anchors, _f), _p(rb, _f), _p(scores, _f), B, N, C,
        ctypes.c_float(image_hw[0]), ctypes.c_float(image_hw[1]), int(bool(do_clip)),
        ctypes.c_float(score_threshold), ctypes.c_float(iou_threshold), int(max_ret),
        _p(ob, _f), _p(os_, _f), _p(ol, _i), _p(ok, _i), _p(on, _i), int(threads))
    if rc != 0:
        raise MemoryError('odbox_oracle_decode_nms')
    return ob, os_, ol, ok, on

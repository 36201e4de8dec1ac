"""CPU oracle (NumPy fp32) for the box-processing hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product path (object_detection_amd/) never does and fails loudly when the
HIP library is missing.

PARITY UNPINNED: the reference (bioothod/object_detection) has no tests, golden vectors
or stored outputs for this path, and TensorFlow is not installed
here, so the reference cannot be executed.  This file is a restatement, op by op, of the
reference's TensorFlow op sequences, each op rounded separately in IEEE binary32 like a
TF CPU kernel, plus a restatement of the third-party TF kernels it calls (ArgMax, OneHot,
Where, TopKV2, NonMaxSuppressionV3 -- TensorFlow, unpinned, API usage dates it to
TF 2.1-2.3) from their published behaviour.  It is checked against the hand-derived
known-answer cases of SURVEY.md §8c (tests/test_oracle_kat.py) and against the
independent C restatement in oracle/odbox_oracle.c.

Every function cites the reference file:line it follows.
"""
import math

import numpy as np

F32 = np.float32
I32 = np.int32


def _f32(x):
    return np.asarray(x, dtype=F32)


# ----------------------------------------------------------------------------------------
# anchors
# ----------------------------------------------------------------------------------------

def base_anchors(size, aspect_ratios):
    """A = 3*len(ratios) anchors centred on the origin, float64 math, rounded to fp32 last.

    Follows anchors.py:89-111 (`AnchorGenerator._generate`) with the hard-coded scales of
    anchors.py:37-40.  Anchor a uses ratio a//3 and scale a%3 (np.repeat / np.tile order).
    """
    scales = [2 ** 0, 2 ** (1 / 3.0), 2 ** (2 / 3.0)]
    out = np.zeros((len(aspect_ratios) * 3, 4), dtype=np.float64)
    for a in range(out.shape[0]):
        ratio = float(aspect_ratios[a // 3])
        side = float(size) * float(scales[a % 3])
        area = side * side
        w = math.sqrt(area / ratio)
        h = w * ratio
        out[a] = (0.0 - w * 0.5, 0.0 - h * 0.5, w - w * 0.5, h - h * 0.5)
    return out.astype(F32)


def tile_anchors(base, fh, fw, stride):
    """anchors.py:60-87: cell k = i*fw + j (y outer, x inner), row k*A + a = base[a] + shift."""
    base = _f32(base)
    sx = (np.arange(fw, dtype=F32) + F32(0.5)) * F32(stride)
    sy = (np.arange(fh, dtype=F32) + F32(0.5)) * F32(stride)
    gx, gy = np.meshgrid(sx, sy)  # 'xy' indexing, like tf.meshgrid
    shifts = np.stack([gx.reshape(-1), gy.reshape(-1), gx.reshape(-1), gy.reshape(-1)], axis=1)
    out = base[None, :, :] + shifts[:, None, :].astype(F32)
    return out.reshape(-1, 4).astype(F32)


ANCHOR_SIZES = (32, 64, 128, 256, 512)       # config.py:34
ANCHOR_STRIDES = (8, 16, 32, 64, 128)        # config.py:35
ANCHOR_RATIOS = (1, 2, .5)                   # config.py:36


def generate_anchors(im_shape, sizes=ANCHOR_SIZES, strides=ANCHOR_STRIDES, ratios=ANCHOR_RATIOS):
    """detection.py:223-236: levels 3..7, feature side im_shape // 2**level, level 3 first."""
    parts = []
    for lvl in range(3, 8):
        side = im_shape // (2 ** lvl)
        parts.append(tile_anchors(base_anchors(sizes[lvl - 3], ratios), side, side, strides[lvl - 3]))
    return np.concatenate(parts, axis=0)


def yolo_shaped_anchors(im_shape=416, sizes=(32, 64, 128), strides=(8, 16, 32)):
    """BASELINE config 2 grid: AnchorGenerator(size, [1], stride) at 3 strides."""
    parts = []
    for size, stride in zip(sizes, strides):
        side = im_shape // stride
        parts.append(tile_anchors(base_anchors(size, [1]), side, side, stride))
    return np.concatenate(parts, axis=0)


# ----------------------------------------------------------------------------------------
# training side: IoU, matching, encoding
# ----------------------------------------------------------------------------------------

def bbox_overlap(bboxes, gt_bboxes):
    """bndbox.py:254-299.  bboxes [B,N,4], gt [B,G,4] xyxy -> iou [B,N,G], -1 for padded GT."""
    bb = _f32(bboxes)
    gt = _f32(gt_bboxes)
    bx0, by0, bx1, by1 = (bb[:, :, k:k + 1] for k in range(4))            # [B,N,1]
    gx0, gy0, gx1, gy1 = (gt[:, None, :, k] for k in range(4))            # [B,1,G]
    ix0 = np.maximum(bx0, gx0)
    ix1 = np.minimum(bx1, gx1)
    iy0 = np.maximum(by0, gy0)
    iy1 = np.minimum(by1, gy1)
    i_area = np.maximum(ix1 - ix0, F32(0)) * np.maximum(iy1 - iy0, F32(0))
    bb_area = (by1 - by0) * (bx1 - bx0)
    gt_area = (gy1 - gy0) * (gx1 - gx0)
    u_area = ((bb_area + gt_area) - i_area) + F32(1e-8)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = (i_area / u_area).astype(F32)
    pad = (gt.max(axis=-1) < F32(0))[:, None, :]                           # [B,1,G]
    return np.where(pad, F32(-1), iou).astype(F32)


def compute_gt_annotations(anchors, annotations, negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:215-274 -> (positive [B,N] bool, ignore [B,N] bool, batched_indices [B*N,2] i32).

    tf.argmax: first maximal index.  No per-GT forced match.
    """
    anchors = _f32(anchors)
    annotations = _f32(annotations)
    B = annotations.shape[0]
    N = anchors.shape[0]
    argmax = np.empty((B, N), dtype=I32)
    maxv = np.empty((B, N), dtype=F32)
    for b in range(B):  # image by image so the [N,G] matrix stays small
        ov = bbox_overlap(anchors[None], annotations[b:b + 1])[0]
        # Eigen's argmax / max reducers start from (0, lowest) and replace on a strict >: a NaN overlap (inf - inf,
        # only possible when a box area overflows binary32) never wins; numpy's would.  Same rule as odbox_oracle.c.
        ov = np.where(np.isnan(ov), F32(-np.inf), ov)
        argmax[b] = ov.argmax(axis=-1).astype(I32)
        maxv[b] = ov.max(axis=-1)
    positive = maxv >= F32(positive_overlap)
    ignore = (maxv > F32(negative_overlap)) & ~positive
    ignore = ignore | (maxv < F32(0))
    bidx = np.repeat(np.arange(B, dtype=I32), N)
    batched = np.stack([bidx, argmax.reshape(-1)], axis=1).astype(I32)
    return positive, ignore, batched, maxv


def gt_best_anchors(anchors, annotations):
    """Per ground-truth row, (argmax, max) of `bbox_overlap` over its anchor axis (first maximal
    anchor).  Not on the reference's live path (anchors.py:253-272 is per-anchor only); its only
    per-GT argmax is the diagnostic `np.argmax(ious)` of coco.py:399-402.  Definition of the extra
    output `out_gt_best` of odbox_match_encode."""
    annotations = np.asarray(annotations, dtype=np.float32)
    B = annotations.shape[0]
    a = np.broadcast_to(np.asarray(anchors, dtype=np.float32), (B,) + np.shape(anchors))
    iou = bbox_overlap(a, annotations)                     # [B,N,G]
    idx = np.argmax(iou, axis=1).astype(np.int32)          # [B,G]
    return idx, np.take_along_axis(iou, idx[:, None, :], axis=1)[:, 0, :]


def bbox_transform(anchors, gt_boxes):
    """anchors.py:277-299: offsets / anchor size, then (t - 0) / 0.2f (a true fp32 division)."""
    a = _f32(anchors)
    g = _f32(gt_boxes)
    aw = a[..., 2] - a[..., 0]
    ah = a[..., 3] - a[..., 1]
    with np.errstate(divide='ignore', invalid='ignore'):
        t = np.stack([(g[..., 0] - a[..., 0]) / aw,
                      (g[..., 1] - a[..., 1]) / ah,
                      (g[..., 2] - a[..., 2]) / aw,
                      (g[..., 3] - a[..., 3]) / ah], axis=-1).astype(F32)
        t = ((t - F32(0)) / F32(0.2)).astype(F32)
    return t


def anchor_targets_bbox(anchors, image_hw, bndboxes, labels, num_classes,
                        negative_overlap=0.4, positive_overlap=0.5):
    """anchors.py:116-213.  Returns dict with the reference outputs `regression` [B,N,5] and
    `labels` [B,N,C+1] (returned by the reference in that order, :213) plus the compact side
    outputs match_idx [B,N] i32, state [B,N] i8 (-1 ignore / 0 bg / 1 fg), max_iou [B,N]."""
    anchors = _f32(anchors)
    bndboxes = _f32(bndboxes)
    labels = np.asarray(labels, dtype=I32)
    B, G = labels.shape
    N = anchors.shape[0]
    h = F32(image_hw[0])
    w = F32(image_hw[1])
    positive, ignore, batched, maxv = compute_gt_annotations(anchors, bndboxes,
                                                             negative_overlap, positive_overlap)
    xc = (anchors[:, 0] + anchors[:, 2]) / F32(2)
    yc = (anchors[:, 1] + anchors[:, 3]) / F32(2)
    out_mask = (xc >= w) | (yc >= h)
    ignore = ignore | out_mask[None, :]
    match = batched[:, 1].reshape(B, N)
    chose_labels = np.take_along_axis(labels, match, axis=1)
    lab = np.where(positive, chose_labels, I32(-1))
    lab = np.where(ignore, I32(-1), lab)
    onehot = np.zeros((B, N, num_classes), dtype=F32)
    ok = (lab >= 0) & (lab < num_classes)     # tf.one_hot: out-of-range index -> all-zero row
    bi, ni = np.nonzero(ok)
    onehot[bi, ni, lab[bi, ni]] = F32(1)
    chose_boxes = np.take_along_axis(bndboxes, match[:, :, None], axis=1)
    regression = bbox_transform(anchors[None], chose_boxes)
    state = np.where(ignore, F32(-1), positive.astype(F32)).astype(F32)
    return {
        'regression': np.concatenate([regression, state[..., None]], axis=-1).astype(F32),
        'labels': np.concatenate([onehot, state[..., None]], axis=-1).astype(F32),
        'match_idx': match.astype(I32),
        'state': state.astype(np.int8),
        'max_iou': maxv,
    }


# ----------------------------------------------------------------------------------------
# inference side: decode, clip, NMS
# ----------------------------------------------------------------------------------------

def regress_bndboxes(boxes, regressors):
    """bndbox.py:71-99: x' = x + (r*0.2f + 0)*size, every op rounded separately."""
    b = _f32(boxes)
    r = _f32(regressors)
    w = b[..., 2] - b[..., 0]
    h = b[..., 3] - b[..., 1]
    s = F32(0.2)
    z = F32(0)
    return np.stack([b[..., 0] + (r[..., 0] * s + z) * w,
                     b[..., 1] + (r[..., 1] * s + z) * h,
                     b[..., 2] + (r[..., 2] * s + z) * w,
                     b[..., 3] + (r[..., 3] * s + z) * h], axis=-1).astype(F32)


def clip_boxes(boxes, im_size):
    """bndbox.py:102-115: x in [0, w-1], y in [0, h-1]; clip_by_value = min(max(v, lo), hi)."""
    b = _f32(boxes)
    h = F32(im_size[0] - 1)
    w = F32(im_size[1] - 1)
    z = F32(0)
    return np.stack([np.minimum(np.maximum(b[..., 0], z), w),
                     np.minimum(np.maximum(b[..., 1], z), h),
                     np.minimum(np.maximum(b[..., 2], z), w),
                     np.minimum(np.maximum(b[..., 3], z), h)], axis=-1).astype(F32)


def scale_boxes(boxes, from_size, to_size):
    """bndbox.py:10-41 (ratio is a Python float, becomes an fp32 constant)."""
    b = _f32(boxes)
    rw = F32(from_size[1] / to_size[1])
    rh = F32(from_size[0] / to_size[0])
    return np.stack([b[:, 0] * rw, b[:, 1] * rh, b[:, 2] * rw, b[:, 3] * rh], axis=1).astype(F32)


def normalize_bndboxes(boxes, image_size):
    """bndbox.py:44-68."""
    b = _f32(boxes)
    h, w = image_size
    dw = F32(w - 1)
    dh = F32(h - 1)
    return np.stack([b[:, 0] / dw, b[:, 1] / dh, b[:, 2] / dw, b[:, 3] / dh], axis=1).astype(F32)


def tf_nms_iou(bi, bj):
    """IoU of TensorFlow's NonMaxSuppressionV3 (third-party kernel, call site bndbox.py:170-175;
    semantics restated in SURVEY.md §8c): corners normalised by min/max, 0 if an area <= 0,
    no epsilon, fp32."""
    y0i, y1i = min(bi[0], bi[2]), max(bi[0], bi[2])
    x0i, x1i = min(bi[1], bi[3]), max(bi[1], bi[3])
    y0j, y1j = min(bj[0], bj[2]), max(bj[0], bj[2])
    x0j, x1j = min(bj[1], bj[3]), max(bj[1], bj[3])
    ai = F32(F32(y1i - y0i) * F32(x1i - x0i))
    aj = F32(F32(y1j - y0j) * F32(x1j - x0j))
    if ai <= 0 or aj <= 0:
        return F32(0)
    ih = max(F32(min(y1i, y1j) - max(y0i, y0j)), F32(0))
    iw = max(F32(min(x1i, x1j) - max(x0i, x0j)), F32(0))
    inter = F32(ih * iw)
    return F32(inter / F32(F32(ai + aj) - inter))


def tf_non_max_suppression(boxes, scores, max_output_size, iou_threshold, score_threshold):
    """tf.image.non_max_suppression restated: candidates score > thr, visited
    by (score desc, index asc); kept iff IoU <= thr against every kept box; stop at
    max_output_size.  Returns indices in selection order."""
    boxes = _f32(boxes)
    scores = _f32(scores)
    thr = F32(iou_threshold)
    cand = [i for i in range(scores.shape[0]) if scores[i] > F32(score_threshold)]
    cand.sort(key=lambda i: (-float(scores[i]), i))
    kept = []
    for i in cand:
        if len(kept) >= max_output_size:
            break
        if all(not (tf_nms_iou(boxes[i], boxes[j]) > thr) for j in kept):
            kept.append(i)
    return np.asarray(kept, dtype=I32)


def nms(bboxes, class_scores, score_threshold=0.3, iou_threshold=0.45, max_ret=100):
    """bndbox.py:117-252.  Returns (boxes [B,max_ret,4] xyxy, scores [B,max_ret],
    labels [B,max_ret] i32, keep_anchor [B,max_ret] i32 with -1 padding, count [B] i32).
    keep_anchor/count are oracle-side extras used to check the keep-list bit-exactly."""
    bboxes = _f32(bboxes)
    class_scores = _f32(class_scores)
    B, N, C = class_scores.shape
    out_b = np.zeros((B, max_ret, 4), dtype=F32)
    out_s = np.zeros((B, max_ret), dtype=F32)
    out_l = np.zeros((B, max_ret), dtype=I32)
    out_k = np.full((B, max_ret), -1, dtype=I32)
    out_n = np.zeros((B,), dtype=I32)
    for b in range(B):
        labels = class_scores[b].argmax(axis=-1).astype(I32)      # first max
        scores = class_scores[b].max(axis=-1)
        yxyx = bboxes[b][:, [1, 0, 3, 2]]
        keep = np.nonzero(scores > F32(score_threshold))[0]        # ascending anchor order
        s_boxes, s_scores, s_labels = yxyx[keep], scores[keep], labels[keep]
        cat_boxes, cat_scores, cat_labels, cat_anchor = [], [], [], []
        for c in range(C):
            ci = np.nonzero(s_labels == c)[0]
            if ci.size == 0:
                continue
            sel = tf_non_max_suppression(s_boxes[ci], s_scores[ci], max_ret,
                                         iou_threshold, score_threshold)
            for k in sel:
                cat_boxes.append(s_boxes[ci[k]])
                cat_scores.append(s_scores[ci[k]])
                cat_labels.append(c)
                cat_anchor.append(keep[ci[k]])
        n = len(cat_scores)
        if n == 0:
            continue
        k = min(max_ret, n)
        # tf.math.top_k(sorted=True): descending, equal values -> lower index first
        order = sorted(range(n), key=lambda i: (-float(cat_scores[i]), i))[:k]
        for r, i in enumerate(order):
            y0, x0, y1, x1 = cat_boxes[i]
            out_b[b, r] = (x0, y0, x1, y1)
            out_s[b, r] = cat_scores[i]
            out_l[b, r] = cat_labels[i]
            out_k[b, r] = cat_anchor[i]
        out_n[b] = k
    return out_b, out_s, out_l, out_k, out_n


def decode_and_nms(anchors, regressors, class_scores, image_hw,
                   score_threshold=0.3, iou_threshold=0.45, max_ret=100):
    """encoder.py:200-214: tile anchors, regress -> clip -> nms."""
    anchors = _f32(anchors)
    regressors = _f32(regressors)
    boxes = regress_bndboxes(np.broadcast_to(anchors[None], regressors.shape), regressors)
    boxes = clip_boxes(boxes, image_hw)
    return nms(boxes, class_scores, score_threshold, iou_threshold, max_ret)


# ----------------------------------------------------------------------------------------
# consumer of the targets: focal loss + Huber
# ----------------------------------------------------------------------------------------

def focal_loss_elements(y_true, y_pred, gamma=1.5, alpha=0.25):
    """metric.py:16-30, element by element in fp32 (before the reductions of :31-36)."""
    y = _f32(y_true)
    p = np.minimum(np.maximum(_f32(y_pred), F32(1e-6)), F32(1.0) - F32(1e-6))
    pos = y == F32(1)
    a = np.where(pos, F32(alpha), F32(1) - F32(alpha)).astype(F32)
    pt = np.where(pos, p, F32(1) - p).astype(F32)
    w = np.power(F32(1) - pt, F32(gamma)).astype(F32)
    return ((-a * w) * np.log(pt)).astype(F32)


def softmax(x):
    """tf.nn.softmax over the last axis in fp32 (metric.py:19-20): exp(x - max) / sum."""
    x = _f32(x)
    e = np.exp(x - x.max(axis=-1, keepdims=True)).astype(F32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)


def focal_loss(y_true, y_pred, gamma=1.5, alpha=0.25, reduction='sum', from_logits=False):
    """metric.py:9-38.  The order of TF's reductions is not specified, so the sums are taken in
    float64 and compared with a tolerance."""
    if from_logits:
        y_pred = softmax(y_pred)
    rows = focal_loss_elements(y_true, y_pred, gamma, alpha).sum(axis=-1, dtype=np.float64)
    if reduction == 'mean':
        return rows.mean()
    if reduction == 'sum':
        return rows.sum()
    return rows


def huber(y_true, y_pred, delta=1.0):
    """tf.keras.losses.Huber(delta)(y_true, y_pred) per sample: mean over the last axis."""
    e = (_f32(y_pred) - _f32(y_true)).astype(F32)
    ae = np.abs(e)
    d = F32(delta)
    h = np.where(ae <= d, F32(0.5) * (e * e), d * ae - F32(0.5) * (d * d)).astype(F32)
    return h.mean(axis=-1, dtype=np.float64)


def model_metric(anchors, image_hw, true_bboxes, true_labels, pred_bboxes, pred_scores, num_classes,
                 gamma=1.5, alpha=0.25, delta=1.0, with_grads=False):
    """metric.py:97-123 (ModelMetric.__call__ without the whole-image head): returns
    (dist_loss, class_loss, normalizer[, grad_boxes, grad_scores]); the gradients are those of
    the UNNORMALISED sums w.r.t. the predictions, analytic, in float64."""
    t = anchor_targets_bbox(anchors, image_hw, true_bboxes, true_labels, num_classes)
    labels, regression = t['labels'], t['regression']
    states = labels[:, :, -1]
    not_ignore = states != F32(-1)
    fg = states == F32(1)
    normalizer = float(fg.sum())
    pred_scores = _f32(pred_scores)
    pred_bboxes = _f32(pred_bboxes)
    class_sum = focal_loss(labels[:, :, :-1][not_ignore], pred_scores[not_ignore], gamma, alpha, 'sum')
    dist_sum = huber(regression[:, :, :-1][fg], pred_bboxes[fg], delta).sum()
    dist_loss, class_loss = dist_sum, class_sum
    if normalizer != 0:
        dist_loss, class_loss = dist_sum / normalizer, class_sum / normalizer
    if not with_grads:
        return dist_loss, class_loss, normalizer
    gb = np.zeros(pred_bboxes.shape, dtype=np.float64)
    e = pred_bboxes[fg].astype(np.float64) - regression[:, :, :-1][fg].astype(np.float64)
    gb[fg] = np.where(np.abs(e) <= delta, e, delta * np.sign(e)) / 4.0
    y = labels[:, :, :-1] == F32(1)
    p64 = pred_scores.astype(np.float64)
    eps = float(F32(1e-6))
    hi = float(F32(1.0) - F32(1e-6))
    pc = np.clip(p64, eps, hi)
    a = np.where(y, alpha, 1.0 - alpha)
    pt = np.where(y, pc, 1.0 - pc)
    dpt = a * (gamma * np.power(1.0 - pt, gamma - 1.0) * np.log(pt) - np.power(1.0 - pt, gamma) / pt)
    gs = np.where(y, dpt, -dpt)
    gs = np.where((p64 >= eps) & (p64 <= hi), gs, 0.0)
    gs = np.where(not_ignore[:, :, None], gs, 0.0)
    return dist_loss, class_loss, normalizer, gb, gs


def foreground_accuracy(anchors, image_hw, true_bboxes, true_labels, pred_scores, num_classes):
    """metric.py:128-130: tf.keras.metrics.CategoricalAccuracy over the foreground anchors, i.e.
    argmax(one-hot label row) == argmax(predicted scores) with tf.argmax's first-maximum rule (an
    all-zero one-hot row, tf.one_hot of an out-of-range label, has argmax 0).
    Returns (hits, foreground count); the metric is their ratio."""
    t = anchor_targets_bbox(anchors, image_hw, true_bboxes, true_labels, num_classes)
    labels = t['labels']
    fg = labels[:, :, -1] == F32(1)
    y = labels[:, :, :-1][fg]
    p = _f32(pred_scores)[fg]
    if y.shape[0] == 0:
        return 0, 0
    return int((np.argmax(y, axis=-1) == np.argmax(p, axis=-1)).sum()), int(y.shape[0])


def categorical_accuracy(y_true, y_pred):
    """tf.keras.metrics.CategoricalAccuracy on dense rows (metric.py:136): (hits, rows)."""
    y_true, y_pred = _f32(y_true), _f32(y_pred)
    rows = int(np.prod(y_true.shape[:-1]))
    return int((np.argmax(y_true, axis=-1) == np.argmax(y_pred, axis=-1)).sum()), rows


# ----------------------------------------------------------------------------------------
# ground-truth preparation feeding the matcher (detection.py)
# ----------------------------------------------------------------------------------------

def polygon2bbox(poly, want_yx=False, want_wh=False):
    """detection.py:77-95.  poly [N,4,2] -> [N,4], in poly's dtype (fp32 or fp16)."""
    poly = np.asarray(poly)
    x, y = poly[..., 0], poly[..., 1]
    xmin, ymin = x.min(axis=1, keepdims=True), y.min(axis=1, keepdims=True)
    xmax, ymax = x.max(axis=1, keepdims=True), y.max(axis=1, keepdims=True)
    if want_yx:
        return np.concatenate([ymin, xmin, ymax, xmax], 1)
    if want_wh:
        return np.concatenate([xmin, ymin, (xmax - xmin).astype(poly.dtype), (ymax - ymin).astype(poly.dtype)], 1)
    return np.concatenate([xmin, ymin, xmax, ymax], 1)


def prepare_example_boxes(orig_bboxes, orig_labels, orig_hw, image_size, max_items, dtype=np.float32,
                          want_yx=False, want_wh=False):
    """The box arithmetic of detection.py:97-141 on the evaluation path (no augmentation):
    orig_bboxes [n,4] = (x0, y0, w, h) in original pixels, orig_hw = (height, width) of the image.
    Returns (new_bboxes [min(n,max_items),4] in `dtype`, new_labels)."""
    bb = _f32(orig_bboxes).reshape(-1, 4)
    x0, y0, w, h = (bb[:, k:k + 1] for k in range(4))                       # :98
    H, W = F32(orig_hw[0]), F32(orig_hw[1])                                 # :100-101
    mx = np.maximum(H, W)                                                   # :103
    x0 = x0 + (mx - W) / F32(2)                                             # :110
    y0 = y0 + (mx - H) / F32(2)                                             # :111
    p0 = np.concatenate([x0, y0], 1)                                        # :118-121
    p1 = np.concatenate([x0 + w, y0], 1)
    p2 = np.concatenate([x0 + w, y0 + h], 1)
    p3 = np.concatenate([x0, y0 + h], 1)
    poly = np.stack([p0, p1, p2, p3], 1).astype(F32)                        # :123
    cur = F32(int(mx))                                                      # :104-109,131
    return prepare_polygons(poly, orig_labels, cur, image_size, max_items, dtype, want_yx, want_wh)


def prepare_polygons(word_poly, labels, current_image_size, image_size, max_items, dtype=np.float32,
                     want_yx=False, want_wh=False):
    """detection.py:131-141: rescale the corner polygons [n,4,2] to the network input, cast to the
    image dtype, polygon2bbox, cut at max_items_in_image."""
    poly = _f32(word_poly).reshape(-1, 4, 2)
    poly = ((poly / F32(current_image_size)).astype(F32) * F32(image_size)).astype(F32)   # :135
    poly = poly.astype(dtype)                                                # :136
    boxes = polygon2bbox(poly, want_yx, want_wh)                             # :138
    return boxes[:max_items], np.asarray(labels, dtype=I32)[:max_items]     # :140-141


def padded_batch_boxes(examples, max_items):
    """detection.py:257-262 for the two GT tensors: examples = [(boxes [n_i,4], labels [n_i])],
    padded with -1 to [B,max_items,4] fp32 / [B,max_items] int32."""
    B = len(examples)
    out_b = np.full((B, max_items, 4), -1, dtype=F32)
    out_l = np.full((B, max_items), -1, dtype=I32)
    for b, (bb, ll) in enumerate(examples):
        n = len(ll)
        out_b[b, :n] = np.asarray(bb, dtype=F32)
        out_l[b, :n] = ll
    return out_b, out_l


# ----------------------------------------------------------------------------------------
# output side of inference: detections back to original image pixels (infdet.py)
# ----------------------------------------------------------------------------------------

def restore_detections(coords, scores, orig_hw, image_size):
    """infdet.py:134-174, the arithmetic of run_eval's per-detection loop, literally: Python floats
    (fp64), `break` at the first score <= 0.  coords [B,R,4] f32, scores [B,R], orig_hw [B,2]
    (imh, imw) ints.  Returns (boxes [B,R,4] f64, zero beyond the count; counts [B])."""
    coords = _f32(coords)
    scores = _f32(scores)
    B, R = scores.shape
    out = np.zeros((B, R, 4), dtype=np.float64)
    counts = np.zeros(B, dtype=I32)
    for b in range(B):
        imh, imw = int(orig_hw[b][0]), int(orig_hw[b][1])
        max_side = max(imh, imw)
        pad_y = (max_side - imh) / 2
        pad_x = (max_side - imw) / 2
        square_scale = max_side / image_size
        n = 0
        for coord, score in zip(coords[b], scores[b]):
            if float(score) <= 0:
                break
            scaled_bb = coord.astype(float) * square_scale
            xmin, ymin, xmax, ymax = scaled_bb
            xmin -= pad_x
            xmax -= pad_x
            ymin -= pad_y
            ymax -= pad_y
            out[b, n] = [xmin, ymin, xmax, ymax]
            n += 1
        counts[b] = n
    return out, counts

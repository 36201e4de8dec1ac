"""Drop-in for the loss side of the reference module `metric` (metric.py) on MI355X: the consumer
of the box targets, fused with the matcher so the one-hot `[B,N,C+1]` label tensor and the
`[B,N,5]` regression tensor are never materialised.

`ModelMetric(all_anchors, num_classes)(images, true_bboxes, true_labels, true_image_labels,
pred_bboxes, pred_scores, pred_image_scores, training)` returns
`(dist_loss, class_loss, image_class_loss)` like metric.py:97-138 (device scalars) and keeps the
reference's running statistics in `train_metric` (`Metric`, metric.py:40-73): means of the losses
and the two categorical accuracies, accumulated on the device without a host sync; the foreground
accuracy's numerator comes out of the same kernel launch as the losses (out_loss[3]).
HIP only (odbox_match_loss / odbox_focal_loss), no CPU fallback.
"""
import torch

from . import _lib
from .anchors import _image_hw

def _workspace(dev, need):
    return _lib.workspace('loss', dev, need)


def _focal_launch(lib, y_true, y_pred, gamma, alpha, want_grad):
    n = y_pred.numel()
    out = torch.empty(4, dtype=torch.float32, device=y_pred.device)
    grad = torch.empty_like(y_pred) if want_grad else None
    ws = _workspace(y_pred.device, lib.odbox_focal_loss_workspace(n))
    _lib.check(lib.odbox_focal_loss(_lib.ptr(y_true), _lib.ptr(y_pred), n, float(gamma), float(alpha),
                                    _lib.ptr(out), _lib.ptr(grad), _lib.ptr(ws), ws.numel(),
                                    _lib.stream_ptr()), 'odbox_focal_loss')
    return out[1], grad


class _FocalLossFn(torch.autograd.Function):
    """Differentiable dense focal loss (the whole-image head is [B, num_image_classes]: small, so
    the forward launch writes the gradient and backward only scales it)."""

    @staticmethod
    def forward(ctx, y_pred, y_true, gamma, alpha):
        loss, grad = _focal_launch(_lib.load(), y_true, y_pred, gamma, alpha, True)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None


def _focal_rows_launch(lib, y_true, y_pred, gamma, alpha, from_logits, want_grad):
    C = int(y_pred.shape[-1]) if y_pred.dim() else 1
    rows = y_pred.numel() // max(C, 1)
    out = torch.empty(y_pred.shape[:-1] if y_pred.dim() else (), dtype=torch.float32, device=y_pred.device)
    grad = torch.empty_like(y_pred) if want_grad else None
    _lib.check(lib.odbox_focal_loss_rows(_lib.ptr(y_true), _lib.ptr(y_pred), rows, C, float(gamma), float(alpha),
                                         int(bool(from_logits)), _lib.ptr(out), _lib.ptr(grad), _lib.stream_ptr()),
               'odbox_focal_loss_rows')
    return out, grad


class _FocalRowsFn(torch.autograd.Function):
    """Per-row focal loss (`odbox_focal_loss_rows`), differentiable w.r.t. y_pred -- through the softmax when
    from_logits -- with the gradient written by the forward launch."""

    @staticmethod
    def forward(ctx, y_pred, y_true, gamma, alpha, from_logits):
        rows, grad = _focal_rows_launch(_lib.load(), y_true, y_pred, gamma, alpha, from_logits, True)
        ctx.save_for_backward(grad)
        return rows

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.unsqueeze(-1), None, None, None, None


def focal_loss(y_true, y_pred, gamma: float = 1.5, alpha: float = 0.25, from_logits: bool = False,
               reduction: str = 'sum', return_grad: bool = False):
    """metric.py:9-38 on dense targets.  reduction 'sum' (default, the only one the reference's own callers use),
    'mean' (mean of the per-row sums) or anything else (e.g. None): the per-row sums themselves, shape
    y_pred.shape[:-1], like the reference's fall-through return (:33-38).  from_logits=True applies tf.nn.softmax over
    the last axis first (:19-20).  Differentiable w.r.t. y_pred when it requires grad; return_grad=True instead
    returns d loss / d y_pred explicitly (for an un-reduced loss: of each row's sum w.r.t. that row)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    y_true = _lib.as_dev(y_true, torch.float32, dev).detach()
    y_pred = _lib.as_dev(y_pred, torch.float32, dev)
    if y_true.shape != y_pred.shape:
        raise ValueError('y_true %r and y_pred %r differ in shape' % (tuple(y_true.shape), tuple(y_pred.shape)))
    if from_logits or reduction not in ('sum', 'mean'):
        # row by row: the softmax needs the row, the un-reduced return is one value per row
        if not return_grad and torch.is_grad_enabled() and y_pred.requires_grad:
            rows, grad = _FocalRowsFn.apply(y_pred, y_true, gamma, alpha, from_logits), None
        else:
            rows, grad = _focal_rows_launch(lib, y_true, y_pred.detach(), gamma, alpha, from_logits, return_grad)
        if reduction == 'sum':
            loss = rows.sum()
        elif reduction == 'mean':
            loss = rows.mean()
            grad = grad / max(rows.numel(), 1) if grad is not None else None
        else:
            loss = rows
        return (loss, grad) if return_grad else loss
    n = y_pred.numel()
    scale = 1.0
    if reduction == 'mean':
        rows = max(1, n // max(1, y_pred.shape[-1])) if y_pred.dim() else 1
        scale = 1.0 / rows
    if not return_grad and torch.is_grad_enabled() and y_pred.requires_grad:
        loss = _FocalLossFn.apply(y_pred, y_true, gamma, alpha)
        return loss * scale if scale != 1.0 else loss
    loss, grad = _focal_launch(lib, y_true, y_pred.detach(), gamma, alpha, return_grad)
    if scale != 1.0:
        loss = loss * scale
    if return_grad:
        return loss, (grad * scale if scale != 1.0 else grad)
    return loss


def _pred_dtype(pred_bboxes, pred_scores):
    """fp16 predictions (the reference's --use_fp16 / mixed_float16 run, detection.py:296-300) go to
    the *_f16 entry points as they are; anything else is computed from fp32 tensors."""
    if isinstance(pred_bboxes, torch.Tensor) and isinstance(pred_scores, torch.Tensor):
        for dt in (torch.float16, torch.bfloat16):          # bfloat16: this GPU's native mixed-precision format
            if pred_bboxes.dtype == dt and pred_scores.dtype == dt:
                return dt
    return torch.float32


_SUFFIX = {torch.float32: '', torch.float16: '_f16', torch.bfloat16: '_bf16'}


def _entry(lib, name, dtype):
    return getattr(lib, name + _SUFFIX[dtype])


def _loss_args(lib, anchors, image_hw, gt, lab, B, N, G, C, neg, pos, pb, ps):
    return (_lib.ptr(anchors), _lib.ptr(gt), _lib.ptr(lab), B, N, G, C, float(image_hw[0]), float(image_hw[1]),
            float(neg), float(pos), _lib.ptr(pb), _lib.ptr(ps))


class _DetectionLossFn(torch.autograd.Function):
    """(pred_bboxes, pred_scores) -> (dist_loss, class_loss, npos) with a HIP backward: the forward
    launch computes no gradients; the backward launch (`odbox_match_loss_backward`) re-matches and
    writes both gradients once, already multiplied by upstream_gradient / npos read on the device."""

    @staticmethod
    def forward(ctx, pb, ps, anchors, gt, lab, image_hw, cfg):
        lib = _lib.load()
        B, N, G, C, neg, pos, gamma, alpha, delta = cfg
        out = torch.empty(4, dtype=torch.float32, device=pb.device)
        ws = _workspace(pb.device, lib.odbox_match_loss_workspace(B, N))
        _lib.check(_entry(lib, 'odbox_match_loss', pb.dtype)(
            *_loss_args(lib, anchors, image_hw, gt, lab, B, N, G, C, neg, pos, pb, ps),
            float(gamma), float(alpha), float(delta), _lib.ptr(out), None, None, _lib.ptr(ws), ws.numel(),
            _lib.stream_ptr()), 'odbox_match_loss')
        ctx.save_for_backward(pb, ps, anchors, gt, lab, out)
        ctx.image_hw, ctx.cfg = image_hw, cfg
        dist, cls, npos, hits = out[0], out[1], out[2], out[3]
        ctx.mark_non_differentiable(npos, hits)
        return dist, cls, npos, hits

    @staticmethod
    def backward(ctx, g_dist, g_cls, _g_npos, _g_hits):
        pb, ps, anchors, gt, lab, out = ctx.saved_tensors
        gb, gs = _backward_launch(ctx.cfg, ctx.image_hw, anchors, gt, lab, pb, ps, g_dist, g_cls, out[2:3])
        return gb, gs, None, None, None, None, None


def _scalar(t, dev):
    """An upstream gradient as a 4-byte-aligned fp32 device scalar (None stays None = unused loss)."""
    if t is None:
        return None
    return t.detach().to(device=dev, dtype=torch.float32).reshape(1).contiguous()


def _backward_launch(cfg, image_hw, anchors, gt, lab, pb, ps, up_dist, up_class, normalizer):
    """One `odbox_match_loss_backward` launch: the upstream gradients and the normaliser stay on the
    device, the kernel folds them in; returns (grad_boxes, grad_scores) in the predictions' dtype."""
    lib = _lib.load()
    B, N, G, C, neg, pos, gamma, alpha, delta = cfg
    up_dist, up_class = _scalar(up_dist, pb.device), _scalar(up_class, pb.device)
    gb = torch.empty_like(pb)
    gs = torch.empty_like(ps)
    _lib.check(_entry(lib, 'odbox_match_loss_backward', pb.dtype)(
        *_loss_args(lib, anchors, image_hw, gt, lab, B, N, G, C, neg, pos, pb, ps),
        float(gamma), float(alpha), float(delta), _lib.ptr(up_dist), _lib.ptr(up_class), _lib.ptr(normalizer),
        _lib.ptr(gb), _lib.ptr(gs), _lib.stream_ptr()), 'odbox_match_loss_backward')
    return gb, gs


def detection_loss(anchors, image_hw, true_bboxes, true_labels, pred_bboxes, pred_scores, num_classes,
                   negative_overlap=0.4, positive_overlap=0.5, gamma=1.5, alpha=0.25, huber_delta=1.0,
                   return_grads=False, return_hits=False):
    """The fused matcher + loss (`odbox_match_loss`).  Returns (dist_loss, class_loss, normalizer)
    as device scalars, both losses divided by normalizer = #foreground anchors when it is not 0
    (metric.py:108-123).

    When pred_bboxes / pred_scores require grad (and grad mode is on) the two losses are
    differentiable: `.backward()` runs `odbox_match_loss_backward`.  return_grads=True instead
    returns (grad_boxes [B,N,4], grad_scores [B,N,C]) explicitly: the gradients of the two
    RETURNED losses w.r.t. pred_bboxes / pred_scores (a forward launch and a backward launch that
    reads the normaliser on the device -- cheaper than scaling unnormalised gradients afterwards).

    float16 predictions are read as they are (widened to fp32 in registers like metric.py:16-17);
    their gradients come back in float16.

    return_hits=True appends the number of foreground anchors whose top predicted class is the
    matched class (hits / normalizer = the CategoricalAccuracy of metric.py:128-130)."""
    lib = _lib.load()
    dev = _lib.require_gpu()
    anchors = _lib.as_dev(anchors, torch.float32, dev)
    gt = _lib.as_dev(true_bboxes, torch.float32, dev)
    lab = _lib.as_dev(true_labels, torch.int32, dev)
    pdt = _pred_dtype(pred_bboxes, pred_scores)
    pb = _lib.as_dev(pred_bboxes, pdt, dev)
    ps = _lib.as_dev(pred_scores, pdt, dev)
    B, G = int(gt.shape[0]), int(gt.shape[1])
    N, C = int(anchors.shape[0]), int(num_classes)
    if tuple(lab.shape) != (B, G) or tuple(pb.shape) != (B, N, 4) or tuple(ps.shape) != (B, N, C):
        raise ValueError('shape mismatch: labels %r, pred_bboxes %r, pred_scores %r for B=%d N=%d G=%d C=%d'
                         % (tuple(lab.shape), tuple(pb.shape), tuple(ps.shape), B, N, G, C))
    if not return_grads and torch.is_grad_enabled() and (pb.requires_grad or ps.requires_grad):
        cfg = (B, N, G, C, negative_overlap, positive_overlap, gamma, alpha, huber_delta)
        res = _DetectionLossFn.apply(pb, ps, anchors.detach(), gt.detach(), lab, (float(image_hw[0]), float(image_hw[1])), cfg)
        return res if return_hits else res[:3]
    pb, ps = pb.detach(), ps.detach()
    out = torch.empty(4, dtype=torch.float32, device=dev)
    ws = _workspace(dev, lib.odbox_match_loss_workspace(B, N))
    _lib.check(_entry(lib, 'odbox_match_loss', pdt)(
        *_loss_args(lib, anchors, image_hw, gt, lab, B, N, G, C, negative_overlap, positive_overlap, pb, ps),
        float(gamma), float(alpha), float(huber_delta), _lib.ptr(out), None, None,
        _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), 'odbox_match_loss')
    res = (out[0], out[1], out[2])
    if return_grads:
        cfg = (B, N, G, C, negative_overlap, positive_overlap, gamma, alpha, huber_delta)
        one = torch.ones(1, dtype=torch.float32, device=dev)
        res = res + _backward_launch(cfg, image_hw, anchors, gt, lab, pb, ps, one, one, out[2:3])
    return res + (out[3],) if return_hits else res


class _Mean:
    """tf.keras.metrics.Mean on the device: update_state never syncs, result() does."""

    def __init__(self):
        self.total = None
        self.count = None

    def reset_states(self):
        self.total = None
        self.count = None

    def update_state(self, value, weight=1.0):
        value = torch.as_tensor(value, dtype=torch.float32).detach()
        weight = torch.as_tensor(weight, dtype=torch.float32, device=value.device).detach()
        if self.total is None:
            self.total, self.count = value.clone(), weight.clone()
        else:
            self.total += value
            self.count += weight

    def result(self):
        if self.total is None:
            return 0.0
        c = float(self.count)
        return float(self.total) / c if c != 0 else 0.0


class Metric:
    """metric.py:40-73: the running statistics the reference prints per epoch.  The means take one
    value per step; the accuracies are hits / samples over all steps (CategoricalAccuracy)."""

    def __init__(self, training=True, **kwargs):
        self.training = training
        self.total_loss = _Mean()
        self.reg_loss = _Mean()
        self.dist_loss = _Mean()
        self.ce_loss = _Mean()
        self.ce_image_loss = _Mean()
        self.ce_acc = _Mean()
        self.ce_image_acc = _Mean()

    def reset_states(self):
        for m in (self.total_loss, self.reg_loss, self.dist_loss, self.ce_loss, self.ce_image_loss, self.ce_acc,
                  self.ce_image_acc):
            m.reset_states()

    def str_result(self):
        return 'total_loss: {:.4f}, reg_loss: {:.3f}, dist: {:.3f}, ce: {:.3f}, image_ce: {:.3f}, acc: {:.3f}, image_acc: {:.3f}'.format(
            self.total_loss.result(), self.reg_loss.result(), self.dist_loss.result(), self.ce_loss.result(),
            self.ce_image_loss.result(), self.ce_acc.result(), self.ce_image_acc.result())


class ModelMetric:
    """metric.py:76-138.  `dtype` is accepted for signature parity (it only casts the image
    height / width in the reference)."""

    def __init__(self, all_anchors, num_classes: int, dtype=torch.float32, **kwargs):
        self.all_anchors = all_anchors
        self.num_classes = num_classes
        self.dtype = dtype
        self.train_metric = Metric(training=True, name='train_metric')

    def str_result(self, training):
        return self.train_metric.str_result()

    def reset_states(self):
        self.train_metric.reset_states()

    def __call__(self, images, true_bboxes, true_labels, true_image_labels, pred_bboxes, pred_scores,
                 pred_image_scores, training):
        dist_loss, class_loss, npos, hits = detection_loss(self.all_anchors, _image_hw(images), true_bboxes,
                                                           true_labels, pred_bboxes, pred_scores, self.num_classes,
                                                           return_hits=True)
        m = self.train_metric
        m.dist_loss.update_state(dist_loss)                                   # metric.py:125-126
        m.ce_loss.update_state(class_loss)
        m.ce_acc.update_state(hits, npos)                                     # :128-130
        image_class_loss = 0.
        if pred_image_scores is not None:
            image_class_loss = focal_loss(true_image_labels, pred_image_scores, reduction='sum')
            m.ce_image_loss.update_state(image_class_loss)                    # :134-136
            dev = image_class_loss.device
            yt = torch.as_tensor(true_image_labels, device=dev)
            yp = torch.as_tensor(pred_image_scores, device=dev).detach()
            m.ce_image_acc.update_state((yt.argmax(-1) == yp.argmax(-1)).sum(), float(yt[..., 0].numel()))
        return dist_loss, class_loss, image_class_loss

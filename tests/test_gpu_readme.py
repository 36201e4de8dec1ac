"""The usage snippet of README.md, run as written (small random data)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_readme_snippet_runs():
    from object_detection_amd import anchors, bndbox, metric, detection, cocoeval, config
    rng = np.random.default_rng(0)
    B, S, C = 2, 256, 80
    all_anchors = anchors.generate_anchors(config.AnchorsConfig(), S)
    N = all_anchors.shape[0]
    boxes_xywh = [np.array([[30, 40, 60, 50], [100, 90, 40, 80]], np.float32), np.array([[10, 10, 100, 120]], np.float32)]
    labels = [np.array([3, 7], np.int32), np.array([1], np.int32)]
    orig_hw = np.array([(S, S), (S, S)], np.int32)
    true_bboxes, true_labels = detection.prepare_batch(boxes_xywh, labels, orig_hw, S)
    assert tuple(true_bboxes.shape) == (B, 64, 4) and tuple(true_labels.shape) == (B, 64)
    images = np.zeros((B, S, S, 3), np.float32)
    pred_bboxes = torch.as_tensor(rng.normal(0, 0.5, (B, N, 4)).astype(np.float32)).cuda().requires_grad_(True)
    pred_scores = torch.as_tensor(rng.uniform(0.01, 0.2, (B, N, C)).astype(np.float32)).cuda().requires_grad_(True)
    loss = metric.ModelMetric(all_anchors, num_classes=C)
    dist_loss, class_loss, image_loss = loss(images, true_bboxes, true_labels, None, pred_bboxes, pred_scores, None, True)
    (dist_loss + class_loss).backward()
    assert torch.isfinite(pred_scores.grad).all() and pred_bboxes.grad.abs().sum() > 0 and image_loss == 0.
    sc = pred_scores.detach().clone()
    sc[0, 100, 3] = 0.9
    sc[1, 2000, 1] = 0.8
    boxes, scores, det_labels = bndbox.decode_and_nms(all_anchors, pred_bboxes.detach(), sc, (S, S))
    assert int((scores > 0).sum()) == 2
    ev = cocoeval.COCOeval(num_classes=C)
    ev.add_batch(boxes, scores, det_labels, true_bboxes, true_labels)
    ev.accumulate()
    ev.summarize()
    assert len(ev.stats) == 12
    assert 'acc:' in loss.str_result(True)

"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the
NumPy oracle) against the C oracle on the CPU and against the HIP path on the GPU."""
import os

import numpy as np
import pytest

from oracle import c_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return np.load(os.path.join(HERE, 'golden', name))


def test_c_oracle_matches_golden_match_encode():
    g = load('match_encode_s128.npz')
    r = c_oracle.match_encode(g['anchors'], g['image_hw'], g['gt_boxes'], g['gt_labels'], int(g['num_classes']))
    for k in ('match_idx', 'state', 'max_iou', 'labels', 'regression'):
        np.testing.assert_array_equal(r[k], g[k], err_msg=k)
    assert (g['state'] == 1).sum() >= 5 and (g['state'][2] == -1).all()


def test_c_oracle_matches_golden_decode_nms():
    g = load('decode_nms_s128.npz')
    r = c_oracle.decode_nms(g['anchors'], g['regressors'], g['class_scores'], g['image_hw'], max_ret=int(g['max_ret']))
    for got, k in zip(r, ('boxes', 'scores', 'labels', 'keep_anchor', 'count')):
        np.testing.assert_array_equal(got, g[k], err_msg=k)
    assert g['count'].min() >= 10


@pytest.mark.gpu
def test_hip_matches_golden_match_encode():
    from object_detection_amd import anchors as A
    g = load('match_encode_s128.npz')
    r = A.match_encode(g['anchors'], g['image_hw'], g['gt_boxes'], g['gt_labels'], int(g['num_classes']),
                       dense=True, compact=True)
    for k in ('match_idx', 'state', 'max_iou', 'labels'):
        np.testing.assert_array_equal(r[k].cpu().numpy(), g[k], err_msg=k)
    reg = r['regression'].cpu().numpy()
    np.testing.assert_array_equal(reg[..., 4], g['regression'][..., 4])
    np.testing.assert_allclose(reg, g['regression'], rtol=1e-5, atol=0)


@pytest.mark.gpu
def test_hip_matches_golden_decode_nms():
    from object_detection_amd import bndbox as BB
    g = load('decode_nms_s128.npz')
    r = BB.decode_and_nms(g['anchors'], g['regressors'], g['class_scores'], tuple(g['image_hw']),
                          max_ret=int(g['max_ret']), return_keep=True)
    got = [t.cpu().numpy() for t in r]
    np.testing.assert_array_equal(got[3], g['keep_anchor'])
    np.testing.assert_array_equal(got[4], g['count'])
    np.testing.assert_array_equal(got[2], g['labels'])
    np.testing.assert_array_equal(got[1], g['scores'])
    np.testing.assert_allclose(got[0], g['boxes'], rtol=1e-5, atol=0)


# ---- the rows beyond match / NMS: loss, COCO matching, GT preparation, detection restore ----------

def test_oracle_still_reproduces_the_later_fixtures():
    """Guards the oracle itself: a change to oracle/*.py that alters any committed vector shows up
    here, on the CPU, before it can silently move the target of the GPU parity tests."""
    from oracle import coco_oracle as CO
    from oracle import odbox_oracle as O
    g = load('loss_s128.npz')
    C, hw = int(g['num_classes']), tuple(g['image_hw'])
    d, c, n, gb, gs = O.model_metric(g['anchors'], hw, g['gt_boxes'], g['gt_labels'], g['pred_boxes'], g['pred_scores'], C,
                                     with_grads=True)
    assert (d, c, n) == (float(g['dist_loss']), float(g['class_loss']), float(g['normalizer'])) and n > 0
    assert O.foreground_accuracy(g['anchors'], hw, g['gt_boxes'], g['gt_labels'], g['pred_scores'], C)[0] == int(g['hits'])
    np.testing.assert_array_equal(gs.astype(np.float32), g['grad_scores_sum'])
    np.testing.assert_array_equal(gb.astype(np.float32), g['grad_boxes_sum'])
    g = load('coco_eval_small.npz')
    r = CO.dense_eval(g['dt_boxes'], g['dt_scores'], g['dt_labels'], g['gt_boxes'], g['gt_labels'], tuple(g['image_hw']),
                      int(g['num_classes']))
    for k in ('dt_match', 'dt_ignore', 'gt_match', 'gt_ignore', 'dt_rank'):
        np.testing.assert_array_equal(r[k], g[k], err_msg=k)
    assert (g['dt_match'] > 0).any()
    g = load('restore.npz')
    out, cnt = O.restore_detections(g['coords'], g['scores'], g['orig_hw'], int(g['image_size']))
    np.testing.assert_array_equal(out, g['boxes'])
    np.testing.assert_array_equal(cnt, g['counts'])


@pytest.mark.gpu
def test_hip_matches_golden_loss():
    from object_detection_amd import metric as M
    g = load('loss_s128.npz')
    C, hw = int(g['num_classes']), tuple(g['image_hw'])
    d, c, n, gb, gs, hits = M.detection_loss(g['anchors'], hw, g['gt_boxes'], g['gt_labels'], g['pred_boxes'],
                                             g['pred_scores'], C, return_grads=True, return_hits=True)
    assert float(n) == float(g['normalizer']) and float(hits) == int(g['hits'])
    np.testing.assert_allclose([float(d), float(c)], [float(g['dist_loss']), float(g['class_loss'])], rtol=1e-5)
    nrm = float(g['normalizer'])
    np.testing.assert_allclose(gb.cpu().numpy() * nrm, g['grad_boxes_sum'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gs.cpu().numpy() * nrm, g['grad_scores_sum'], rtol=1e-4,
                               atol=1e-6 * np.abs(g['grad_scores_sum']).max())
    np.testing.assert_allclose(float(M.focal_loss(g['image_labels'], g['image_scores'])), float(g['image_focal_sum']), rtol=1e-5)


@pytest.mark.gpu
def test_hip_matches_golden_coco_tables_and_stats():
    from object_detection_amd import cocoeval as CE
    g = load('coco_eval_small.npz')
    C = int(g['num_classes'])
    out = CE.evaluate_img(g['dt_boxes'], g['dt_scores'], g['dt_labels'], g['gt_boxes'], g['gt_labels'], C)
    for k in ('dt_match', 'dt_ignore', 'gt_match', 'gt_ignore', 'dt_rank'):
        np.testing.assert_array_equal(out[k].cpu().numpy(), g[k], err_msg=k)
    ev = CE.COCOeval(C)
    ev.add_batch(g['dt_boxes'], g['dt_scores'], g['dt_labels'], g['gt_boxes'], g['gt_labels'])
    ev.accumulate()
    ev.summarize()
    np.testing.assert_array_equal(np.asarray(ev.stats, np.float64), g['stats'])


@pytest.mark.gpu
def test_hip_matches_golden_gt_prepare_and_restore():
    import torch
    from object_detection_amd import detection as D
    g = load('gt_prepare.npz')
    offs = g['row_offsets']
    boxes = [g['boxes_xywh'][offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    labels = [g['labels'][offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    for name, dt in (('f32', torch.float32), ('f16', torch.float16)):
        bb, ll = D.prepare_batch(boxes, labels, g['orig_hw'], int(g['image_size']), max_items=int(g['max_items']), dtype=dt)
        np.testing.assert_array_equal(bb.float().cpu().numpy(), g['boxes_' + name], err_msg=name)
        np.testing.assert_array_equal(ll.cpu().numpy(), g['labels_' + name], err_msg=name)
    g = load('restore.npz')
    out, cnt = D.restore_detections(g['coords'], g['scores'], g['orig_hw'], int(g['image_size']))
    np.testing.assert_array_equal(out.cpu().numpy(), g['boxes'])
    np.testing.assert_array_equal(cnt.cpu().numpy(), g['counts'])

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

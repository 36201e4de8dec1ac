"""CPU-side checks of the boundary: libodbox.so loads without a GPU and exports exactly the
symbols include/odbox.h declares; the Python drop-in modules keep the reference signatures and
fail loudly (no CPU fallback) when no GPU is present."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

from object_detection_amd import _lib, anchors, bndbox, build, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    build.build()
    return _lib.load()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'odbox.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(odbox_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = header_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string(lib):
    assert lib.odbox_version() == 100
    assert isinstance(lib.odbox_last_error(), bytes)


def test_workspace_query_needs_no_gpu(lib):
    assert lib.odbox_decode_nms_workspace(64, 49104, 90, 100, 0) > 64 * 49104 * 8
    assert lib.odbox_decode_nms_workspace(1, 1 << 21, 90, 100, 0) == 0      # N over the key limit
    assert lib.odbox_decode_nms_workspace(1, 100, 5000, 100, 0) == 0


def test_argument_validation_happens_before_any_gpu_work(lib):
    rc = lib.odbox_match_encode(None, None, None, 1, 1, 1, 1, 1.0, 1.0, 0.4, 0.5, None, None, None, None, None, None)
    assert rc == -1 and b'NULL' in lib.odbox_last_error()


def test_reference_signatures_are_kept():
    def params(f):
        return list(inspect.signature(f).parameters)
    assert params(anchors.anchor_targets_bbox) == ['anchors', 'images', 'bndboxes', 'labels', 'num_classes',
                                                   'negative_overlap', 'positive_overlap', 'dtype']
    assert params(anchors.compute_gt_annotations) == ['anchors', 'annotations', 'negative_overlap', 'positive_overlap']
    assert params(anchors.bbox_transform) == ['anchors', 'gt_boxes']
    assert params(anchors.AnchorGenerator.__init__) == ['self', 'size', 'aspect_ratios', 'stride']
    assert params(anchors.generate_anchors) == ['anchors_config', 'im_shape']
    assert params(bndbox.nms)[:5] == ['bboxes', 'class_scores', 'score_threshold', 'iou_threshold', 'max_ret']
    assert params(bndbox.regress_bndboxes) == ['boxes', 'regressors']
    assert params(bndbox.clip_boxes) == ['boxes', 'im_size']
    assert params(bndbox.bbox_overlap) == ['bboxes', 'gt_bboxes']
    assert params(bndbox.scale_boxes) == ['boxes', 'from_size', 'to_size']
    assert params(bndbox.normalize_bndboxes) == ['boxes', 'image_size']
    sig = inspect.signature(bndbox.nms)
    assert (sig.parameters['score_threshold'].default, sig.parameters['iou_threshold'].default,
            sig.parameters['max_ret'].default) == (0.3, 0.45, 100)
    c = config.AnchorsConfig()
    assert c.sizes == (32, 64, 128, 256, 512) and c.strides == (8, 16, 32, 64, 128) and c.ratios == (1, 2, .5)


def test_base_anchors_are_host_float64_then_fp32():
    from oracle import odbox_oracle as O
    g = anchors.AnchorGenerator(64, (1, 2, .5), 16)
    assert len(g) == 9 and g.anchors.dtype == torch.float32
    np.testing.assert_array_equal(g.anchors.numpy(), O.base_anchors(64, (1, 2, .5)))


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback():
    g = anchors.AnchorGenerator(32, [1], 8)
    with pytest.raises(_lib.OdboxError):
        g((4, 4, 3))
    with pytest.raises(_lib.OdboxError):
        bndbox.nms(np.zeros((1, 4, 4), np.float32), np.zeros((1, 4, 2), np.float32))

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
    assert lib.odbox_version() == 201
    assert isinstance(lib.odbox_last_error(), bytes)


def test_workspace_query_needs_no_gpu(lib):
    assert lib.odbox_decode_nms_workspace(64, 49104, 90, 100, 0) > 64 * 49104 * 8
    assert lib.odbox_decode_nms_workspace(1, 1 << 21, 90, 100, 0) == 0      # N over the key limit
    assert lib.odbox_decode_nms_workspace(1, 100, 5000, 100, 0) == 0


def test_argument_validation_happens_before_any_gpu_work(lib):
    rc = lib.odbox_match_encode(None, None, None, 1, 1, 1, 1, 1.0, 1.0, 0.4, 0.5, None, None, None, None, None, None, None)
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


def test_caller_side_signatures_are_kept():
    """metric.py, detection.py and cocoeval.py: the callers either side of the path."""
    from object_detection_amd import cocoeval, detection, metric

    def params(f):
        return list(inspect.signature(f).parameters)
    assert params(metric.focal_loss)[:6] == ['y_true', 'y_pred', 'gamma', 'alpha', 'from_logits', 'reduction']
    sig = inspect.signature(metric.focal_loss)
    assert (sig.parameters['gamma'].default, sig.parameters['alpha'].default, sig.parameters['reduction'].default) == (1.5, 0.25, 'sum')
    assert params(metric.ModelMetric.__init__)[:3] == ['self', 'all_anchors', 'num_classes']
    assert params(metric.ModelMetric.__call__) == ['self', 'images', 'true_bboxes', 'true_labels', 'true_image_labels',
                                                   'pred_bboxes', 'pred_scores', 'pred_image_scores', 'training']
    assert params(detection.polygon2bbox) == ['poly', 'want_yx', 'want_wh']
    assert detection.MAX_ITEMS_IN_IMAGE == 64 and detection.generate_anchors is anchors.generate_anchors
    p = cocoeval.Params()                                              # cocoeval.py:506-515
    np.testing.assert_array_equal(p.iouThrs, np.linspace(.5, .95, 10))
    np.testing.assert_array_equal(p.recThrs, np.linspace(0, 1, 101))
    assert p.maxDets == [1, 10, 100] and p.areaRngLbl == ['all', 'small', 'medium', 'large']
    assert p.areaRng == [[0, 1e10], [0, 1024], [1024, 9216], [9216, 1e10]]
    for name in ('accumulate', 'summarize', 'add_batch'):
        assert callable(getattr(cocoeval.COCOeval, name))
    ev = cocoeval.COCOeval(3)
    ev.accumulate()                                                    # nothing fed: every entry stays -1
    assert ev.eval['precision'].shape == (10, 101, 3, 4, 3) and (ev.eval['precision'] == -1).all()
    assert (ev.summarize() == -1).all()


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU behaviour')
def test_no_cpu_fallback_in_the_caller_side_modules():
    from object_detection_amd import cocoeval, detection, metric
    z = np.zeros
    with pytest.raises(_lib.OdboxError):
        metric.focal_loss(z((2, 3), np.float32), z((2, 3), np.float32))
    with pytest.raises(_lib.OdboxError):
        metric.detection_loss(z((4, 4), np.float32), (8, 8), z((1, 1, 4), np.float32), z((1, 1), np.int32),
                              z((1, 4, 4), np.float32), z((1, 4, 2), np.float32), 2)
    with pytest.raises(_lib.OdboxError):
        detection.prepare_batch([z((1, 4))], [z(1)], [(8, 8)], 16)
    with pytest.raises(_lib.OdboxError):
        cocoeval.evaluate_img(z((1, 2, 4)), z((1, 2)), z((1, 2)), z((1, 1, 4)), z((1, 1)), 2)
    with pytest.raises(_lib.OdboxError):
        cocoeval.compute_iou(z((1, 2, 4)), z((1, 1, 4)))


def test_oracle_is_not_imported_by_the_product():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, 'object_detection_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'oracle/' not in text.replace('oracle/_', ''), f


def test_every_entry_rejects_bad_arguments_without_touching_the_gpu(lib):
    """Return-code convention of include/odbox.h: ODBOX_EINVAL (-1) / ODBOX_ENOSPC (-3) with a
    message, decided on the host before any HIP call -- so this runs without a GPU."""
    import ctypes
    buf = (ctypes.c_float * 64)()                        # a host buffer: never dereferenced by these calls
    p = ctypes.cast(buf, ctypes.c_void_p)
    odd = ctypes.c_void_p(p.value + 4)                   # 4-byte aligned only
    thr = (ctypes.c_double * 2)(0.5, 0.75)
    rng = (ctypes.c_double * 2)(0.0, 1e10)
    cases = [
        ('odbox_anchor_tile', (None, 9, 4, 4, 8.0, p, None), -1),
        ('odbox_anchor_tile', (p, -1, 4, 4, 8.0, p, None), -1),
        ('odbox_match_encode', (p, p, p, 1, 4, 2, 0, 8.0, 8.0, 0.4, 0.5, p, p, None, None, None, None, None), -1),      # C < 1
        ('odbox_match_encode', (odd, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, None, None, None, None, None), -1),    # alignment
        ('odbox_match_encode', (p, p, None, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, None, None, None, None, None), -1),   # labels needed
        ('odbox_match_encode', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, None, None, None, odd, None), -1),   # out_gt_best 8-byte aligned
        ('odbox_match_loss', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, p, None, p, 1 << 20, None), -1),   # one gradient only
        ('odbox_match_loss', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, None, None, p, 4, None), -3),        # workspace
        ('odbox_match_loss_noalloc', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, None, None, None, None), -1),      # no out_loss
        ('odbox_match_loss_noalloc', (p, p, p, 4096, 49104, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, None, None, None), -3),  # over the scratch limit
        ('odbox_match_loss_backward', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, p, p, None, p, None), -1),    # no grad_boxes
        ('odbox_match_loss_backward', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, p, p, odd, p, None), -1),     # alignment
        ('odbox_match_loss_backward', (p, p, p, 1, 4, 2, 0, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, None, None, None, p, p, None), -1),  # C < 1
        ('odbox_match_loss_f16', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, odd, p, 1.5, 0.25, 1.0, p, None, None, p, 1 << 20, None), -1),   # boxes 8-byte aligned
        ('odbox_match_loss_f16', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, None, None, None, p, 1 << 20, None), -1),  # no out_loss
        ('odbox_match_loss_backward_f16', (p, p, p, 1, 4, 2, 2, 8.0, 8.0, 0.4, 0.5, p, p, 1.5, 0.25, 1.0, p, p, p, p, None, None), -1),
        ('odbox_focal_loss', (p, p, 16, 1.5, 0.25, None, None, p, 1 << 20, None), -1),
        ('odbox_focal_loss', (p, p, 16, 1.5, 0.25, p, None, p, 1, None), -3),
        ('odbox_bbox_overlap', (None, p, 1, 4, 2, 0, p, None), -1),
        ('odbox_bbox_transform', (p, 0, None, 1, 4, p, None), -1),
        ('odbox_regress_boxes', (p, 0, p, 1, 4, None, None), -1),
        ('odbox_clip_boxes', (None, 4, 8.0, 8.0, p, None), -1),
        ('odbox_scale_boxes', (p, 4, 1.0, 1.0, 0, None, None), -1),
        ('odbox_decode_nms', (p, p, p, 1, 4, 5000, 8.0, 8.0, 1, 0.3, 0.45, 100, 0, p, p, p, None, None, p, 1 << 30, None), -1),   # C over the key limit
        ('odbox_decode_nms', (p, p, p, 1, 4, 2, 8.0, 8.0, 1, 0.3, 0.45, 100, 0, p, p, p, None, None, p, 16, None), -3),
        ('odbox_decode_nms', (p, p, p, 1, 4, 2, 8.0, 8.0, 1, 0.3, 0.45, 100, -1, p, p, p, None, None, p, 1 << 30, None), -1),
        ('odbox_decode_nms_mixed', (p, p, 1, p, 1, 1, 4, 3, 8.0, 8.0, 1, 0.3, 0.45, 100, 0, p, p, p, None, None, p, 1 << 30, None), -1),   # 16-bit scores, odd C
        ('odbox_decode_nms_mixed', (p, p, 3, p, 1, 1, 4, 2, 8.0, 8.0, 1, 0.3, 0.45, 100, 0, p, p, p, None, None, p, 1 << 30, None), -1),   # unknown dtype
        ('odbox_coco_records', (None, 4, p, None, None, None), -1),
        ('odbox_coco_iou', (p, None, 1, 2, 2, p, None), -1),
        ('odbox_coco_match', (p, p, p, p, p, 1, 2, 2, 3, thr, 2, rng, 1, 100, p, p, None, None, p, p, 4, None), -3),
        ('odbox_coco_match', (p, p, p, p, p, 1, 2, 2, 3, thr, 99, rng, 1, 100, p, p, None, None, p, p, 1 << 20, None), -1),     # too many thresholds
        ('odbox_coco_match', (p, p, p, p, p, 1, 2, 5000, 3, thr, 2, rng, 1, 100, p, p, None, None, p, p, 1 << 20, None), -1),   # G over the limit
        ('odbox_gt_prepare_xywh', (p, p, None, 8, p, 2, 4, 512.0, 0, 0, p, p, None), -1),
        ('odbox_gt_prepare_xywh', (p, None, p, 8, p, 2, 4, 512.0, 0, 0, p, p, None), -1),                                          # labels NULL with rows > 0
        ('odbox_gt_prepare_poly', (None, p, p, 8, p, 2, 4, 512.0, 0, 0, p, p, None), -1),                                          # polygons NULL with rows > 0
        ('odbox_restore_boxes', (p, p, None, 2, 4, 512, p, p, None), -1),
        ('odbox_pack_records', (p, p, None, 8, p, None), -1),
        ('odbox_unpack_records', (p, 8, odd, p, p, None), -1),
        ('odbox_restore_boxes', (p, p, p, 2, 4, 0, p, p, None), -1),                                                             # image_size
        ('odbox_gt_prepare_poly', (p, p, p, 8, p, 2, 4, 512.0, 0, 7, p, p, None), -1),                                              # mode
        ('odbox_gt_prepare_xywh', (p, p, p, 8, p, 2, 4, 0.0, 0, 0, p, p, None), -1),                                                # image_size
        ('odbox_legacy_iou', (p, p, None, 2, 4, 0, p, None), -1),                                                                # mode 0 needs the areas
        ('odbox_legacy_iou', (p, p, p, 2, 4, 3, p, None), -1),                                                                   # mode
        ('odbox_legacy_iou', (p, odd, p, 2, 4, 1, p, None), -1),                                                                 # alignment
        ('odbox_legacy_match', (p, p, None, p, 1, 4, 2, 0.5, p, p, None, None, None), -1),                                       # GT rows NULL with G > 0
        ('odbox_legacy_match', (p, p, p, p, 70000, 4, 2, 0.5, p, p, None, None, None), -1),                                      # B over the grid limit
    ]
    for name, args, want in cases:
        rc = getattr(lib, name)(*args)
        assert rc == want, (name, rc, lib.odbox_last_error())
        assert name.encode() in lib.odbox_last_error(), (name, lib.odbox_last_error())
    # the output policy is a process-wide setting: unknown values are refused, a valid call returns the previous one
    assert lib.odbox_set_output_policy(7) == -1 and b'odbox_set_output_policy' in lib.odbox_last_error()
    assert lib.odbox_set_nms_tier(0) == 8 and lib.odbox_set_nms_tier(8) == 0             # default 8, 0 = off
    assert lib.odbox_set_nms_tier(-1) == -1 and b'odbox_set_nms_tier' in lib.odbox_last_error()
    assert lib.odbox_set_nms_fused_tail(0) == 1 and lib.odbox_set_nms_fused_tail(1) == 0  # default on
    assert lib.odbox_set_nms_fused_tail(2) == -1 and b'odbox_set_nms_fused_tail' in lib.odbox_last_error()
    assert lib.odbox_set_output_policy(2) == 0 and lib.odbox_set_output_policy(0) == 2
    # which kernel a dense match+encode takes is host logic (odbox_match_encode_plan: 0 tiles, 1 streamed tiles, 2 sweep)
    assert lib.odbox_match_encode_plan(64, 10647, 20, 80, 1, 0) == 2               # AUTO, a label-stream shape: the sweep (AUTO is a function of the shape)
    assert lib.odbox_match_encode_plan(64, 10647, 20, 80, 0, 0) == 0 and lib.odbox_match_encode_plan(64, 10647, 65, 80, 1, 0) == 0
    assert lib.odbox_match_encode_last_plan() == -1                                # no call yet on this thread
    assert lib.odbox_match_encode_plan(160, 10647, 20, 80, 1, 0) == 2              # 552 MB: streamed, wide rows, short GT list
    assert lib.odbox_match_encode_plan(32, 196416, 20, 80, 1, 0) == 1              # 2 GB: streamed tiles
    assert lib.odbox_match_encode_plan(32, 196416, 81, 80, 1, 0) == 0              # ... but not with a long GT list
    assert lib.odbox_match_encode_plan(128, 49104, 500, 1, 1, 0) == 0
    assert lib.odbox_set_output_policy(2) == 0
    assert [lib.odbox_match_encode_plan(64, 10647, 20, 80, a, g) for a, g in ((1, 0), (0, 0), (1, 1))] == [2, 1, 1]
    assert lib.odbox_match_encode_plan(64, 10647, 65, 80, 1, 0) == 1 and lib.odbox_match_encode_plan(64, 10647, 20, 67, 1, 0) == 1
    assert lib.odbox_match_encode_plan(128, 49104, 500, 1, 1, 0) == 0              # narrow label rows never stream
    assert lib.odbox_match_encode_plan(600, 10647, 20, 20, 1, 0) == 0 and lib.odbox_match_encode_plan(600, 10647, 20, 63, 1, 0) == 1
    assert lib.odbox_match_encode_plan(8192, 300, 3, 80, 1, 0) == 2 and lib.odbox_match_encode_plan(8193, 300, 3, 80, 1, 0) == 1
    assert lib.odbox_match_encode_plan(8192, 100, 64, 80, 1, 0) == 1               # staging the batch's GT rows per block would cost more than the labels
    assert lib.odbox_set_output_policy(0) == 2
    assert lib.odbox_match_encode_plan(0, 1, 1, 1, 1, 0) == -1 and b'odbox_match_encode_plan' in lib.odbox_last_error()
    # empty problems are successes, decided before any pointer is looked at
    assert lib.odbox_anchor_tile(None, 9, 0, 4, 8.0, None, None) == 0
    assert lib.odbox_clip_boxes(None, 0, 8.0, 8.0, None, None) == 0
    assert lib.odbox_decode_nms(None, None, None, 0, 4, 2, 8.0, 8.0, 1, 0.3, 0.45, 100, 0, None, None, None, None, None, None, 0, None) == 0
    assert lib.odbox_coco_match(None, None, None, None, None, 0, 2, 2, 3, thr, 2, rng, 1, 100, None, None, None, None, None, None, 0, None) == 0
    assert lib.odbox_gt_prepare_xywh(None, None, None, 0, None, 0, 4, 512.0, 0, 0, None, None, None) == 0
    assert lib.odbox_legacy_iou(None, None, None, 0, 4, 0, None, None) == 0
    assert lib.odbox_legacy_match(None, None, None, None, 0, 4, 2, 0.5, None, None, None, None, None) == 0

"""GPU parity of the per-GT best anchor (`out_gt_best` of odbox_match_encode; SURVEY.md §0 bullet 3,
north_star's "per-GT best-anchor" reduction): for every ground-truth row the anchor with the largest
IoU over all anchors and that IoU, against the oracle's column argmax of `bbox_overlap` -- index
work, bit-exact, IoU value included.  It is an EXTRA output: asking for it must not change one bit
of the reference's outputs (the reference forces no label from it, anchors.py:253-272)."""
import numpy as np
import pytest
import torch

from object_detection_amd import anchors as A
from object_detection_amd import workloads
from oracle import c_oracle
from oracle import odbox_oracle as O

pytestmark = pytest.mark.gpu


def cpu(t):
    return t.cpu().numpy()


def _check(anchors, gt, ref=None):
    ba, bi = A.gt_best_anchors(anchors, gt)
    ra, ri = O.gt_best_anchors(anchors, gt) if ref is None else ref
    np.testing.assert_array_equal(cpu(ba), ra)
    np.testing.assert_array_equal(cpu(bi).view(np.uint32), np.asarray(ri, np.float32).view(np.uint32))   # the bits, sign of zero included
    return ra, ri


def test_small_grid_with_padding_far_rows_and_duplicates():
    rng = np.random.default_rng(8)
    anchors = O.generate_anchors(256)
    gts = [workloads.gt_boxes(rng, 256, 12, 5, pad_to=16) for _ in range(3)]
    gt = np.stack([g[0] for g in gts])
    gt[0, 0] = -1                                    # a padded row in front: (anchor 0, -1)
    gt[1, 2] = [1000, 1000, 1010, 1010]              # overlaps nothing: every IoU +0 -> anchor 0
    gt[2, 5] = anchors[777]                          # exactly an anchor: IoU 1 / (1 + 1e-8 / area)
    gt[2, 6] = gt[2, 5]                              # the same box twice: two rows, same answer
    ra, ri = _check(anchors, gt)
    assert ra[0, 0] == 0 and ri[0, 0] == -1 and ra[1, 2] == 0 and ri[1, 2] == 0 and ra[2, 5] == 777 == ra[2, 6]


def test_ties_go_to_the_lowest_anchor():
    base = O.generate_anchors(128)
    anchors = np.concatenate([base[:300], base[:300], base[300:]])       # anchors 300..599 repeat 0..299: exact ties
    rng = np.random.default_rng(3)
    gts = [workloads.gt_boxes(rng, 128, 10, 3) for _ in range(2)]
    gt = np.stack([g[0] for g in gts])
    gt[0, 0] = base[10]
    gt[0, 1] = base[299]
    ra, _ = _check(anchors, gt)
    assert ra[0, 0] == 10 and ra[0, 1] == 299
    # rows in different 256-anchor tiles and waves tie as well: the whole array doubled
    _check(np.concatenate([base, base]), gt)


def test_degenerate_rows_and_anchors():
    """Negative-area / zero-area GT rows and anchors give IoU values <= 0 that the zero culling must not hide."""
    rng = np.random.default_rng(77)
    anchors = O.generate_anchors(256)
    inv = anchors.copy()
    inv[::7] = inv[::7][:, [2, 1, 0, 3]]             # x0 > x1: negative width
    inv[5::11, 3] = inv[5::11, 1]                    # zero height
    gts = [workloads.gt_boxes(rng, 256, 20, 5, pad_to=24) for _ in range(3)]
    gt = np.stack([g[0] for g in gts])
    gt[1, 3] = gt[1, 3][[2, 1, 0, 3]]                # inverted GT row
    gt[1, 4, 2] = gt[1, 4, 0]                        # zero-width GT row
    gt[2, :20] += 4000                               # all rows far away
    for a in (anchors, inv):
        ba, bi = A.gt_best_anchors(a, gt)
        ra, ri = O.gt_best_anchors(a, gt)
        np.testing.assert_array_equal(cpu(ba), ra)
        np.testing.assert_array_equal(cpu(bi), ri)   # == : -0 and +0 compare equal, like np.argmax ties them


def test_extra_output_changes_no_other_bit():
    anchors = O.yolo_shaped_anchors(416)
    gt, lab = workloads.match_inputs(2, batch=8)
    plain = A.match_encode(anchors, (416, 416), gt, lab, 80, dense=True, compact=True)
    extra = A.match_encode(anchors, (416, 416), gt, lab, 80, dense=True, compact=True, gt_best=True)
    for k in plain:
        assert torch.equal(plain[k], extra[k]), k
    ra, ri = O.gt_best_anchors(anchors, gt)
    np.testing.assert_array_equal(cpu(extra['gt_best_anchor']), ra)
    np.testing.assert_array_equal(cpu(extra['gt_best_iou']), ri)


def test_cfg1_full_size():
    anchors = O.generate_anchors(512)
    gt, _ = workloads.match_inputs(1)
    _check(anchors, gt, c_oracle.gt_best(anchors, gt, threads=8))


def test_cfg2_full_size_and_padded_to_64():
    anchors = O.yolo_shaped_anchors(416)
    gt, _ = workloads.match_inputs(2)
    _check(anchors, gt, c_oracle.gt_best(anchors, gt, threads=8))
    gt, _ = workloads.match_inputs(2, batch=4, pad_to=64)
    _check(anchors, gt)


def test_cfg4_full_batch():
    """Dense crowd at its full per-GPU size (128 images x 500 GT rows x 49 104 anchors): 3.1 G pairs for the C oracle."""
    anchors = O.generate_anchors(512)
    gt, _ = workloads.match_inputs(4)
    ra, ri = _check(anchors, gt, c_oracle.gt_best(anchors, gt, threads=16))
    assert (ri > 0.25).mean() > 0.5 and len(np.unique(ra)) > 1000


def test_more_rows_than_one_lds_chunk():
    anchors = O.generate_anchors(128)
    rng = np.random.default_rng(12)
    gt = np.stack([workloads.crowd_boxes(rng, 128, 700)[0] for _ in range(2)])      # G = 700 > 512
    _check(anchors, gt)


def test_zero_keys_posted_early_do_not_hide_later_matches():
    """A wave of inverted anchors (negative area: it cannot cull, so it posts its IoU of -0 / +0 with EVERY row) sits in
    the tile that is launched FIRST; the only anchors that overlap the ground truth sit in a tile launched after the
    first resident blocks have finished, so their block starts from the keys that wave left.  A key of the value 0
    (-0 is folded into +0) must read as "nothing to beat" there, and the late tile's anchors must win."""
    import math
    tiles = 4096
    N = tiles * 256
    mult = int(tiles * 0.6180339887) | 1
    while math.gcd(mult, tiles) != 1:
        mult += 2
    first_tile = ((tiles - 1) * mult) % tiles                       # blockIdx 0 (see match_encode_kernel, GTBEST order)
    late_tile = ((tiles - 1 - 3600) * mult) % tiles                 # blockIdx 3600: behind 1 536 resident blocks twice over
    rng = np.random.default_rng(5)
    F = np.float32
    xy = rng.uniform(20000, 30000, (N, 2)).astype(F)                # everybody far away from the ground truth
    wh = rng.uniform(8, 64, (N, 2)).astype(F)
    anchors = np.concatenate([xy, xy + wh], 1)
    anchors[first_tile * 256 + 64: first_tile * 256 + 128] = [1000.0, 0.0, 0.0, 1000.0]     # a whole wave with x0 > x1: area -1e6, unions < 0, IoU -0
    gt = np.array([[[10, 10, 50, 50], [100, 100, 180, 160], [300, 20, 340, 90], [5, 200, 60, 260]]], F)
    for g in range(4):                                              # near-matches in the late tile, two per row
        x0, y0, x1, y1 = gt[0, g]
        anchors[late_tile * 256 + 10 * g + 1] = [x0 + 2, y0 + 1, x1 + 3, y1 - 2]
        anchors[late_tile * 256 + 10 * g + 70] = [x0 - 6, y0 + 4, x1 - 1, y1 + 5]
    ra, ri = _check(anchors, gt, c_oracle.gt_best(anchors, gt, threads=8))
    assert (ri[0] > 0.5).all() and (ra[0] // 256 == late_tile).all()

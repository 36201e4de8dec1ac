"""GPU parity of the training-side path (anchor tiling, IoU matching, target encoding) against
the CPU oracle, through the C-ABI (ctypes) as bound by object_detection_amd.anchors.
Bars (BASELINE.json north_star): match indices, states and one-hot labels bit-exact; encoded
offsets within 1e-5 relative (they are in fact compared bit-exact first and only fall back to
the tolerance if the division differs)."""
import numpy as np
import pytest
import torch

from object_detection_amd import anchors as A
from object_detection_amd import bndbox as BB
from object_detection_amd import config, workloads
from oracle import c_oracle
from oracle import odbox_oracle as O

pytestmark = pytest.mark.gpu
F = np.float32
RTOL = 1e-5


def cpu(t):
    return t.detach().cpu().numpy()


def check_targets(anchors_np, hw, gt, lab, C, neg=0.4, pos=0.5):
    got = A.match_encode(anchors_np, hw, gt, lab, C, neg, pos, dense=True, compact=True)
    ref = c_oracle.match_encode(anchors_np, hw, gt, lab, C, neg, pos, threads=8)
    np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
    np.testing.assert_array_equal(cpu(got['state']), ref['state'])
    np.testing.assert_array_equal(cpu(got['max_iou']), ref['max_iou'])
    np.testing.assert_array_equal(cpu(got['labels']), ref['labels'])
    reg = cpu(got['regression'])
    np.testing.assert_array_equal(reg[..., 4], ref['regression'][..., 4])
    with np.errstate(invalid='ignore'):
        np.testing.assert_allclose(reg, ref['regression'], rtol=RTOL, atol=0, equal_nan=True)
    return got, ref


@pytest.mark.parametrize('S', [512, 1024, 416 + 96])
def test_generate_anchors_bit_exact(S):
    got = cpu(A.generate_anchors(config.AnchorsConfig(), S))
    np.testing.assert_array_equal(got, O.generate_anchors(S))


def test_anchor_generator_api():
    g = A.AnchorGenerator(32, [1], 8)
    assert len(g) == 3 and g.anchors.shape == (3, 4)
    np.testing.assert_array_equal(g.anchors.numpy(), O.base_anchors(32, [1]))
    t = cpu(g((5, 7, 3)))
    np.testing.assert_array_equal(t, O.tile_anchors(O.base_anchors(32, [1]), 5, 7, 8))
    assert cpu(g.tile_anchors_over_feature_map((0, 4))).shape == (0, 4)


def test_known_answers_on_gpu():
    anchors = F([[0, 0, 10, 10]])
    for gt_row, state in (([0, 0, 10, 5], 1), ([0, 0, 10, 4], 0), ([0, 0, 10, 4.5], -1)):
        reg, lab = A.anchor_targets_bbox(anchors, (1, 512, 512, 3), F([[gt_row]]), np.int32([[2]]), 3)
        assert cpu(lab)[0, 0, 3] == state and cpu(reg)[0, 0, 4] == state
        assert cpu(lab)[0, 0, :3].tolist() == ([0, 0, 1] if state == 1 else [0, 0, 0])
    reg, lab = A.anchor_targets_bbox(F([[0, 0, 10, 20]]), (1, 64, 64, 3), F([[[1, 2, 11, 18]]]), np.int32([[0]]), 1)
    np.testing.assert_allclose(cpu(reg)[0, 0, :4], [0.5, 0.5, 0.5, -0.5], rtol=1e-6)
    # out-of-image centre -> ignored even at IoU 1
    reg, lab = A.anchor_targets_bbox(F([[500, 10, 530, 40]]), (1, 512, 512, 3), F([[[500, 10, 530, 40]]]),
                                     np.int32([[0]]), 1)
    assert cpu(lab)[0, 0].tolist() == [0, -1]


@pytest.mark.parametrize('B,N,G,C', [(1, 1, 1, 1), (2, 255, 3, 1), (3, 257, 7, 2), (2, 1000, 20, 80),
                                     (1, 513, 64, 3), (2, 300, 600, 5), (1, 4099, 1030, 90)])
def test_random_shapes_bit_exact(B, N, G, C):
    rng = np.random.default_rng(B * 1000 + N + G + C)
    S = 256
    cx, cy = rng.uniform(0, S, (2, N))
    w, h = rng.uniform(4, S / 2, (2, N))
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(F)
    gx, gy = rng.uniform(0, S, (2, B, G))
    gw, gh = rng.uniform(4, S / 2, (2, B, G))
    gt = np.stack([gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2], -1).astype(F)
    k = min(N, G, 4)
    gt[:, :k] = anchors[:k]                      # exact matches
    if G > 2:
        gt[:, 2] = gt[:, 0]                      # duplicate row -> argmax tie
    lab = rng.integers(0, C, (B, G)).astype(np.int32)
    if G > 4:
        gt[:, -2:] = -1
        lab[:, -2:] = -1
    check_targets(anchors, (S, S - 40), gt, lab, C)


def test_all_padded_and_out_of_range_labels():
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 6, 6, 8)
    gt = np.full((2, 5, 4), -1, F)
    lab = np.full((2, 5), -1, np.int32)
    gt[1, 0] = anchors[10]
    lab[1, 0] = 7                                # >= num_classes: tf.one_hot gives a zero row
    got, ref = check_targets(anchors, (48, 48), gt, lab, 4)
    assert (cpu(got['state'])[0] == -1).all() and (cpu(got['match_idx'])[0] == 0).all()
    assert cpu(got['labels'])[1, 10].tolist() == [0, 0, 0, 0, 1]


def test_padded_64_rows_20_valid():
    S, C = 512, 80
    anchors = O.generate_anchors(S)
    gt, lab = workloads.match_inputs(1, batch=2, pad_to=64)
    assert gt.shape == (2, 64, 4)
    check_targets(anchors, (S, S), gt, lab, C)


def test_cfg1_full_size():
    anchors = O.generate_anchors(512)
    gt, lab = workloads.match_inputs(1)
    got, ref = check_targets(anchors, (512, 512), gt, lab, 80)
    assert (ref['state'] == 1).sum() > 0


def test_cfg2_full_size_yolo_grid():
    anchors = O.yolo_shaped_anchors(416)
    gt, lab = workloads.match_inputs(2)
    check_targets(anchors, (416, 416), gt, lab, 80)


def test_cfg4_dense_crowd_slice():
    anchors = O.generate_anchors(512)
    gt, lab = workloads.match_inputs(4, batch=4)
    assert gt.shape == (4, 500, 4)
    check_targets(anchors, (512, 512), gt, lab, 1)


def test_cfg4_full_batch():
    """cfg 4 at its full per-GPU size (B=128, 500 GT rows, 49 104 anchors): the C oracle needs ~4 s."""
    anchors = O.generate_anchors(512)
    gt, lab = workloads.match_inputs(4)
    assert gt.shape == (128, 500, 4)
    got = A.match_encode(anchors, (512, 512), gt, lab, 1, dense=True, compact=True)
    ref = c_oracle.match_encode(anchors, (512, 512), gt, lab, 1, threads=8, dense=True)
    np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
    np.testing.assert_array_equal(cpu(got['state']), ref['state'])
    np.testing.assert_array_equal(cpu(got['labels']), ref['labels'])
    np.testing.assert_allclose(cpu(got['regression']), ref['regression'], rtol=1e-5, atol=1e-6)
    assert (ref['state'] == 1).sum() > 1000 and (ref['state'] == -1).sum() > 1000


def test_cfg3_large_properties():
    """cfg 3 at full per-GPU size (B=32, N=196416): compact outputs vs the oracle, dense outputs
    through size-independent properties (row sums / state column / one-hot position)."""
    anchors = O.generate_anchors(1024)
    gt, lab = workloads.match_inputs(3, batch=32)
    got = A.match_encode(anchors, (1024, 1024), gt, lab, 80, dense=True, compact=True)
    ref = c_oracle.match_encode(anchors, (1024, 1024), gt, lab, 80, threads=8, dense=False)
    midx, state = got['match_idx'], got['state']
    np.testing.assert_array_equal(cpu(midx), ref['match_idx'])
    np.testing.assert_array_equal(cpu(state), ref['state'])
    labels, reg = got['labels'], got['regression']
    st = state.to(torch.float32)
    assert torch.equal(labels[..., 80], st) and torch.equal(reg[..., 4], st)
    assert torch.equal(labels[..., :80].sum(-1), (state == 1).to(torch.float32))
    want_cls = torch.gather(torch.as_tensor(lab).cuda().long(), 1, midx.long())
    pos = state == 1
    assert torch.equal(labels[..., :80].argmax(-1)[pos], want_cls[pos])


def test_degenerate_boxes_and_gt_culling_stay_exact():
    """The kernel leaves GT rows that cannot touch a wave's anchors out of its loop and folds them
    back in as "first zero row" / "first padded row"; that shortcut is only taken for non-negative
    areas.  Inverted (negative-area) and zero-area anchors / GT rows, far-away GT rows, leading
    padded rows and all-zero-IoU images must still give the oracle's argmax bit for bit."""
    rng = np.random.default_rng(77)
    anchors = O.generate_anchors(256)
    N = anchors.shape[0]
    inv = anchors.copy()
    inv[::7] = inv[::7][:, [2, 1, 0, 3]]             # x0 > x1: negative width
    inv[5::11, 3] = inv[5::11, 1]                    # zero height
    G = 40
    gts = [workloads.gt_boxes(rng, 256, G - 8, 5, pad_to=G) for _ in range(4)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    gt[0, [0, 1]] = -1                               # padding first: argmax must still be row 0 when all IoU <= -1 .. 0
    lab[0, [0, 1]] = -1
    gt[1, 3] = gt[1, 3][[2, 1, 0, 3]]                # inverted GT row
    gt[1, 4, 2] = gt[1, 4, 0]                        # zero-width GT row
    gt[2, :G - 8] = [1000, 1000, 1010, 1010]         # every valid row far outside the image: all IoU 0
    gt[3, :G - 8, :] += 4000                         # all rows far away -> everything culled
    for a in (anchors, inv):
        check_targets(a, (256, 256), gt, lab, 5)


def test_compute_gt_annotations_and_overlap_and_transform():
    rng = np.random.default_rng(5)
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 9, 9, 8)
    gts = [workloads.gt_boxes(rng, 72, 6, 3, pad_to=8) for _ in range(3)]
    gt = np.stack([g[0] for g in gts])
    pos, ign, idx = A.compute_gt_annotations(anchors, gt)
    rpos, rign, ridx, _ = O.compute_gt_annotations(anchors, gt)
    np.testing.assert_array_equal(cpu(pos), rpos)
    np.testing.assert_array_equal(cpu(ign), rign)
    np.testing.assert_array_equal(cpu(idx), ridx)
    assert idx.dtype == torch.int32 and pos.dtype == torch.bool
    iou = BB.bbox_overlap(np.broadcast_to(anchors, (3,) + anchors.shape).copy(), gt)
    np.testing.assert_array_equal(cpu(iou), O.bbox_overlap(np.broadcast_to(anchors, (3,) + anchors.shape), gt))
    np.testing.assert_array_equal(cpu(BB.bbox_overlap(anchors, gt)), cpu(iou))
    chosen = np.take_along_axis(gt, ridx[:, 1].reshape(3, -1, 1), axis=1)
    t = A.bbox_transform(anchors, chosen)
    np.testing.assert_allclose(cpu(t), O.bbox_transform(anchors[None], chosen), rtol=RTOL)


def test_fp16_gt_is_cast_like_the_reference():
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 4, 4, 8)
    gt16 = torch.tensor([[[3.5, 2.25, 30.0, 28.0]]], dtype=torch.float16)
    reg, lab = A.anchor_targets_bbox(anchors, (1, 32, 32, 3), gt16, torch.tensor([[1]]), 2)
    ref = c_oracle.match_encode(anchors, (32, 32), gt16.float().numpy(), np.int32([[1]]), 2)
    np.testing.assert_array_equal(cpu(lab), ref['labels'])


def test_bad_arguments_raise():
    from object_detection_amd._lib import OdboxError
    with pytest.raises(ValueError):
        A.anchor_targets_bbox(np.zeros((4, 3), F), (1, 8, 8, 3), np.zeros((1, 2, 4), F), np.zeros((1, 2), np.int32), 2)
    with pytest.raises(OdboxError):
        A.match_encode(np.zeros((4, 4), F), (8, 8), np.zeros((1, 2, 4), F), np.zeros((1, 2), np.int32), 5000)


@pytest.mark.parametrize('C,N,off', [(1, 257, 1), (2, 300, 2), (3, 511, 3), (5, 1000, 1), (80, 333, 1), (81, 777, 3), (90, 1025, 2)])
def test_unaligned_dense_outputs_and_odd_shapes(C, N, off):
    """The dense outputs are written with 16-byte stores over the flat element range plus scalar
    head / tail stores: output tensors that start 4, 8 or 12 bytes into an allocation, odd anchor
    counts and odd row lengths (C + 1) must land exactly like the oracle's."""
    from object_detection_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda')
    rng = np.random.default_rng(C * 1000 + N)
    B, G, S = 3, 7, 200
    anchors = O.tile_anchors(O.base_anchors(32, (1, 2, .5)), 12, 12, 16)[:N] if N <= 1296 else None
    gts = [workloads.gt_boxes(rng, S, G, C, pad_to=G + 2) for _ in range(B)]
    gt = np.stack([g[0] for g in gts])
    lab = np.stack([g[1] for g in gts])
    a = torch.as_tensor(anchors).to(dev)
    g_ = torch.as_tensor(gt).to(dev)
    l_ = torch.as_tensor(lab).to(dev)
    reg_buf = torch.full((B * N * 5 + 8,), 7.0, device=dev)
    lab_buf = torch.full((B * N * (C + 1) + 8,), 7.0, device=dev)
    reg = reg_buf[off:off + B * N * 5]
    out = lab_buf[off:off + B * N * (C + 1)]
    rc = lib.odbox_match_encode(a.data_ptr(), g_.data_ptr(), l_.data_ptr(), B, N, G + 2, C, float(S), float(S), 0.4, 0.5,
                                reg.data_ptr(), out.data_ptr(), None, None, None, None, _lib.stream_ptr())
    assert rc == 0, lib.odbox_last_error()
    ref = c_oracle.match_encode(anchors, (S, S), gt, lab, C, threads=4)
    np.testing.assert_array_equal(cpu(out).reshape(B, N, C + 1), ref['labels'])
    with np.errstate(invalid='ignore'):
        np.testing.assert_allclose(cpu(reg).reshape(B, N, 5), ref['regression'], rtol=RTOL, atol=0, equal_nan=True)
    # nothing outside the views was touched
    assert float(reg_buf[:off].min()) == 7.0 and float(reg_buf[off + B * N * 5:].min()) == 7.0
    assert float(lab_buf[:off].min()) == 7.0 and float(lab_buf[off + B * N * (C + 1):].min()) == 7.0


def test_streaming_variant_with_off_grid_tiles():
    """Dense labels beyond 512 MB take the cache-bypassing variant (128-anchor tiles, nontemporal
    zero stream, per-anchor patches); on the config-2 grid (N = 10 647) no tile starts on a cache
    line.  B = 160 -> 552 MB of labels, compared with the C oracle element for element."""
    anchors = O.yolo_shaped_anchors(416)
    B = 160
    gt, lab = workloads.match_inputs(2, batch=B)
    got = A.match_encode(anchors, (416, 416), gt, lab, 80, dense=True, compact=True)
    ref = c_oracle.match_encode(anchors, (416, 416), gt, lab, 80, threads=8)
    assert got['labels'].numel() * 4 > 512 << 20
    np.testing.assert_array_equal(cpu(got['state']), ref['state'])
    np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
    assert np.array_equal(cpu(got['labels']), ref['labels'])
    reg = cpu(got['regression'])
    np.testing.assert_array_equal(reg[..., 4], ref['regression'][..., 4])
    np.testing.assert_allclose(reg[..., :4], ref['regression'][..., :4], rtol=1e-5, atol=0)


def test_output_policy_changes_no_bit():
    """odbox_set_output_policy only picks the store path (through / around the caches, 256- or 64-anchor tiles)."""
    anchors = O.yolo_shaped_anchors(416)
    gt, lab = workloads.match_inputs(2, batch=6, pad_to=24)
    outs = {}
    try:
        for pol in ('auto', 'cache', 'stream'):
            A.set_output_policy(pol)
            outs[pol] = A.match_encode(anchors, (416, 416), gt, lab, 80, dense=True, compact=True, gt_best=True)
    finally:
        assert A.set_output_policy('auto') == 'stream'
    ref = c_oracle.match_encode(anchors, (416, 416), gt, lab, 80, threads=4)
    np.testing.assert_array_equal(cpu(outs['stream']['labels']), ref['labels'])
    for pol in ('cache', 'stream'):
        for k in outs['auto']:
            assert torch.equal(outs['auto'][k], outs[pol][k]), (pol, k)
    with pytest.raises(ValueError):
        A.set_output_policy('fast')


def test_ties_and_near_ties_in_the_division_free_matcher():
    """The matcher compares i * rcp(u) and takes the exact IEEE divisions only for a row within 2^-19 of the
    running best.  Rows that tie exactly (duplicates: the FIRST must win), rows one or a few ulps of a
    coordinate apart (IoU values 0 .. a few ulps apart, in both orders) and rows that differ clearly
    must give the oracle's argmax and its max IoU bit for bit."""
    rng = np.random.default_rng(2024)
    anchors = O.generate_anchors(256)
    F = np.float32
    rows = []
    for _ in range(6):
        b0 = workloads.gt_boxes(rng, 256, 1, 3)[0][0]
        up = b0.copy(); up[2] = np.nextafter(up[2], F(1e9))              # one ulp wider
        dn = b0.copy(); dn[3] = np.nextafter(dn[3], F(-1e9))             # one ulp shorter
        far = b0.copy(); far[2] = np.nextafter(np.nextafter(np.nextafter(far[2], F(1e9)), F(1e9)), F(1e9))
        sh = b0 + F(2.0 ** -12)                                          # shifted by a sub-pixel amount
        rows += [b0, up, b0, dn, far, sh, dn, up]
    gt = np.stack([np.stack(rows), np.stack(rows[::-1]), np.stack(rows[3:] + rows[:3])]).astype(F)
    lab = (np.arange(gt.shape[0] * gt.shape[1]).reshape(gt.shape[:2]) % 3).astype(np.int32)
    got = A.match_encode(anchors, (256, 256), gt, lab, 3, dense=False, compact=True)
    ref = c_oracle.match_encode(anchors, (256, 256), gt, lab, 3, threads=4, dense=False)
    np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
    np.testing.assert_array_equal(cpu(got['state']), ref['state'])
    np.testing.assert_array_equal(cpu(got['max_iou']).view(np.uint32), ref['max_iou'].view(np.uint32))
    # the planted rows really are ties / near-ties for many anchors: the two best IoUs of an anchor are within 4 ulps
    iou = O.bbox_overlap(np.broadcast_to(anchors, (3,) + anchors.shape), gt)
    top2 = np.sort(iou, axis=2)[:, :, -2:]
    close = (top2[..., 1] > 0) & (np.abs(top2[..., 1].view(np.int32) - top2[..., 0].view(np.int32)) <= 4)
    assert close.sum() > 1000
    # more rows than one LDS chunk, ties across the chunk boundary
    big = np.concatenate([gt[:1, :40]] * 14, axis=1)[:, :530]
    lab2 = np.zeros(big.shape[:2], np.int32)
    got = A.match_encode(anchors, (256, 256), big, lab2, 1, dense=False, compact=True)
    ref = c_oracle.match_encode(anchors, (256, 256), big, lab2, 1, threads=4, dense=False)
    np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
    np.testing.assert_array_equal(cpu(got['max_iou']).view(np.uint32), ref['max_iou'].view(np.uint32))


def test_many_gt_rows_take_the_tile_major_order_and_tiny_boxes_keep_the_epsilon():
    """More than 64 GT rows into a small output run the tiles of ALL images coarse-first (tile-major order: B >= 8,
    here 9 images and a ragged last tile), and the division-free matcher sums the union as ((a + 1e-8) + g) - i.  Boxes
    of 1e-4 pixels make the 1e-8 a visible part of every union, zero-area anchors and rows included; argmax, state
    and the max IoU must be the oracle's bit for bit at both scales."""
    rng = np.random.default_rng(77)
    F = np.float32
    base = O.generate_anchors(128)[:3001]
    for scale in (1.0, 1e-4 / 32.0):
        anchors = (base * F(scale)).astype(F)
        anchors[5] = anchors[4]                                          # equal anchors
        anchors[17, 2:] = anchors[17, :2]                                # a zero-area anchor
        gt = (np.stack([workloads.gt_boxes(rng, 128, 70, 3)[0] for _ in range(9)]) * F(scale)).astype(F)
        gt[:, 7] = gt[:, 3]                                              # duplicates: the first must win
        gt[:, 9, 2:] = gt[:, 9, :2]                                      # a zero-area row
        gt[2, 60:] = -1.0                                                # padded tail in one image
        lab = rng.integers(0, 3, gt.shape[:2]).astype(np.int32)
        size = (128 * scale, 128 * scale)
        got = A.match_encode(anchors, size, gt, lab, 3, dense=True, compact=True)
        assert A.match_encode_last_plan() == 'tiles'
        ref = c_oracle.match_encode(anchors, size, gt, lab, 3, threads=4, dense=True)
        np.testing.assert_array_equal(cpu(got['match_idx']), ref['match_idx'])
        np.testing.assert_array_equal(cpu(got['state']), ref['state'])
        np.testing.assert_array_equal(cpu(got['max_iou']).view(np.uint32), ref['max_iou'].view(np.uint32))
        np.testing.assert_array_equal(cpu(got['regression']), ref['regression'])          # (NaN offsets of the zero-area anchor compare equal)
        np.testing.assert_array_equal(cpu(got['labels']), ref['labels'])

"""Seeded random shapes through the C-ABI against the C oracle: every combination of odd anchor
counts, tiny and ragged tiles, class counts around the vector width, output tensors that start
anywhere in a cache line, optional outputs, padded / duplicate / degenerate GT rows.  The dense
label tensor is written as a zero stream plus per-anchor patches (match_encode_kernel), so the
buffers are pre-filled with NaN and a guard band around every view is checked."""
import ctypes

import numpy as np
import pytest
import torch

from object_detection_amd import _lib
from oracle import c_oracle

pytestmark = pytest.mark.gpu


def _case(rng):
    B = int(rng.integers(1, 5))
    N = int(rng.choice([1, 2, 3, 31, 63, 64, 65, 127, 255, 256, 257, 300, 511, 513, 777, 1025]))
    G = int(rng.choice([1, 2, 3, 7, 20, 64, 65, 130, 515]))
    C = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 80, 90, 255]))
    S = float(rng.choice([64, 100, 416]))
    cx, cy = rng.uniform(0, S, N), rng.uniform(0, S, N)
    w, h = rng.uniform(2, S / 2, N), rng.uniform(2, S / 2, N)
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    gx, gy = rng.uniform(0, S, (B, G)), rng.uniform(0, S, (B, G))
    gw, gh = rng.uniform(1, S / 2, (B, G)), rng.uniform(1, S / 2, (B, G))
    gt = np.stack([gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2], -1).astype(np.float32)
    lab = rng.integers(-1, C + 2, (B, G)).astype(np.int32)            # includes out-of-range labels
    for b in range(B):
        k = int(rng.integers(0, G + 1))                                # k valid rows, the rest padding
        gt[b, k:] = -1
        lab[b, k:] = -1
        if k >= 2 and rng.random() < 0.5:
            gt[b, 1] = gt[b, 0]                                        # duplicate row: argmax tie
        if k >= 1 and N >= 2 and rng.random() < 0.5:
            gt[b, 0] = anchors[int(rng.integers(0, N))]                # exact match (IoU 1)
        if k >= 3 and rng.random() < 0.3:
            gt[b, 2, 2:] = gt[b, 2, :2]                                # zero-area row
    hw = (S, float(rng.choice([S, S - 16])))
    return anchors, gt, lab, B, N, G, C, hw


@pytest.mark.parametrize('seed', range(48))
def test_match_encode_random_shapes_and_placements(seed):
    rng = np.random.default_rng(7000 + seed)
    anchors, gt, lab, B, N, G, C, hw = _case(rng)
    ref = c_oracle.match_encode(anchors, hw, gt, lab, C)
    lib = _lib.load()
    dev = torch.device('cuda')
    want_reg, want_lab, want_compact = rng.random() < 0.85, rng.random() < 0.85, rng.random() < 0.5
    off_r, off_l = int(rng.integers(0, 40)), int(rng.integers(0, 40))  # floats: anywhere in a 128-byte line and beyond
    guard = 64
    breg = torch.full((guard + off_r + B * N * 5 + guard,), float('nan'), device=dev)
    blab = torch.full((guard + off_l + B * N * (C + 1) + guard,), float('nan'), device=dev)
    reg = breg[guard + off_r: guard + off_r + B * N * 5]
    labels = blab[guard + off_l: guard + off_l + B * N * (C + 1)]
    idx = torch.full((B, N), -7, dtype=torch.int32, device=dev)
    state = torch.full((B, N), -7, dtype=torch.int8, device=dev)
    iou = torch.full((B, N), float('nan'), device=dev)
    a, g, l = (torch.as_tensor(x).to(dev) for x in (anchors, gt, lab))
    p = lambda t, on=True: ctypes.c_void_p(t.data_ptr()) if on else None      # noqa: E731
    rc = lib.odbox_match_encode(p(a), p(g), p(l), B, N, G, C, hw[0], hw[1], 0.4, 0.5, p(reg, want_reg), p(labels, want_lab),
                                p(idx, want_compact), p(state, want_compact), p(iou, want_compact), None, _lib.stream_ptr())
    assert rc == 0, lib.odbox_last_error()
    torch.cuda.synchronize()
    if want_lab:
        np.testing.assert_array_equal(labels.view(B, N, C + 1).cpu().numpy(), ref['labels'])
    else:
        assert torch.isnan(labels).all()
    if want_reg:
        got = reg.view(B, N, 5).cpu().numpy()
        np.testing.assert_array_equal(got[..., 4], ref['regression'][..., 4])
        with np.errstate(invalid='ignore'):
            np.testing.assert_allclose(got[..., :4], ref['regression'][..., :4], rtol=1e-5, atol=0)
    else:
        assert torch.isnan(reg).all()
    if want_compact:
        np.testing.assert_array_equal(idx.cpu().numpy(), ref['match_idx'])
        np.testing.assert_array_equal(state.cpu().numpy(), ref['state'])
        np.testing.assert_array_equal(iou.cpu().numpy(), ref['max_iou'])
    # nothing outside the views was touched
    for buf, off, n in ((breg, off_r, B * N * 5), (blab, off_l, B * N * (C + 1))):
        assert torch.isnan(buf[:guard + off]).all() and torch.isnan(buf[guard + off + n:]).all()


@pytest.mark.parametrize('seed', range(24))
def test_match_loss_random_shapes(seed):
    """Losses, foreground count, accuracy hits and gradients of the fused loss (fp32, fp16 and
    bfloat16 predictions, autograd backward with unequal upstream gradients) against the NumPy oracle."""
    from object_detection_amd import metric as M
    from oracle import odbox_oracle as O
    rng = np.random.default_rng(8000 + seed)
    anchors, gt, lab, B, N, G, C, hw = _case(rng)
    dt = (torch.float32, torch.float16, torch.bfloat16)[seed % 3]
    half = dt != torch.float32
    dev = torch.device('cuda')
    pb = rng.normal(0, 1.5, (B, N, 4)).astype(np.float32)
    ps = rng.uniform(0, 1, (B, N, C)).astype(np.float32)
    ps[rng.random(ps.shape) < 0.02] = 0.0                      # clipped scores carry no gradient
    ps[rng.random(ps.shape) < 0.02] = 1.0
    off = int(rng.integers(0, 9))
    flat = torch.zeros(ps.size + 16, device=dev, dtype=dt)
    tps = flat[off:off + ps.size].view(ps.shape)
    tps.copy_(torch.as_tensor(ps))
    tpb = torch.as_tensor(pb, device=dev).to(dt)
    pb_ref, ps_ref = tpb.float().cpu().numpy(), tps.float().cpu().numpy()
    rd, rc, rn, rgb, rgs = O.model_metric(anchors, hw, gt, lab, pb_ref, ps_ref, C, with_grads=True)
    rh, _ = O.foreground_accuracy(anchors, hw, gt, lab, ps_ref, C)
    tpb = tpb.detach().requires_grad_(True)
    tps = tps.detach().requires_grad_(True)
    d, c, n, hits = M.detection_loss(anchors, hw, gt, lab, tpb, tps, C, return_hits=True)
    assert float(n) == rn and float(hits) == rh
    np.testing.assert_allclose([float(d.detach()), float(c.detach())], [rd, rc], rtol=1e-5, atol=1e-30)
    (3.0 * d + 0.25 * c).backward()
    nrm = rn if rn else 1.0
    floor = 1e-6 * max(1e-30, np.abs(rgs).max() / nrm)
    rt, at = {torch.float32: (1e-4, floor), torch.float16: (1.0 / 1024, max(6e-8, floor)),
              torch.bfloat16: (1.0 / 128, floor)}[dt]
    np.testing.assert_allclose(tpb.grad.float().cpu().numpy(), 3.0 * rgb / nrm, rtol=rt, atol=at if half else 1e-7)
    np.testing.assert_allclose(tps.grad.float().cpu().numpy(), 0.25 * rgs / nrm, rtol=rt, atol=at)


@pytest.mark.parametrize('seed', range(24))
def test_decode_nms_random_shapes(seed):
    """decode + clip + per-class NMS + top-k on random sizes, class counts (every scan geometry),
    thresholds and max_ret against the C oracle: keep-lists, labels, scores exact; boxes 1e-5."""
    from object_detection_amd import bndbox as BB
    rng = np.random.default_rng(9000 + seed)
    B = int(rng.integers(1, 4))
    N = int(rng.choice([1, 5, 64, 100, 511, 1000, 2049, 4097]))
    C = int(rng.choice([1, 2, 3, 5, 8, 20, 64, 90, 130, 300]))
    S = 256.0
    cx, cy = rng.uniform(0, S, N), rng.uniform(0, S, N)
    w, h = rng.uniform(8, 96, N), rng.uniform(8, 96, N)
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    reg = np.clip(rng.normal(0, 0.5, (B, N, 4)), -2, 2).astype(np.float32)
    sc = rng.uniform(0, 0.3, (B, N, C)).astype(np.float32)
    hot = rng.random((B, N)) < rng.choice([0.05, 0.3, 0.9])
    cls = rng.integers(0, min(C, int(rng.choice([1, 3, C]))), (B, N))
    val = (np.round((0.3 + 0.7 * rng.random((B, N))) * 64) / 64).astype(np.float32)     # exact score ties
    bi, ni = np.nonzero(hot)
    sc[bi, ni, cls[bi, ni]] = val[bi, ni]
    thr_s = float(rng.choice([0.3, 0.5]))
    thr_i = float(rng.choice([0.3, 0.45, 0.7]))
    max_ret = int(rng.choice([1, 10, 100, 150]))
    got = BB.decode_and_nms(anchors, reg, sc, (S, S), score_threshold=thr_s, iou_threshold=thr_i, max_ret=max_ret,
                            return_keep=True)
    ref = c_oracle.decode_nms(anchors, reg, sc, (S, S), score_threshold=thr_s, iou_threshold=thr_i, max_ret=max_ret)
    b, s, l, k, n = (t.cpu().numpy() for t in got)
    np.testing.assert_array_equal(n, ref[4])
    np.testing.assert_array_equal(k, ref[3])
    np.testing.assert_array_equal(l, ref[2])
    np.testing.assert_array_equal(s, ref[1])
    np.testing.assert_allclose(b, ref[0], rtol=1e-5, atol=0)


@pytest.mark.parametrize('seed', range(32))
def test_fused_pass_one_block_random_shapes(seed):
    """The fused pass-1 block of decode+NMS (tier_tail_kernel) on random batches made for it: candidate counts below,
    around and far above the tier's rungs, classes from one to hundreds with skewed shares (segments from singletons
    to whole tiers, cut by window boundaries anywhere), groups of identical boxes with adjacent scores (low survival
    rates: second rung, pass 2), exact score ties at the tier's edge, per-class caps from 1 up.  Against the C oracle
    and, bit for bit, against the separate partition / NMS / final kernels."""
    import torch
    from object_detection_amd import bndbox as BB
    rng = np.random.default_rng(77000 + seed)
    B = int(rng.integers(1, 6))
    N = int(rng.choice([300, 1000, 2500, 6000, 12000]))
    C = int(rng.choice([1, 2, 5, 17, 90, 400]))
    max_ret = int(rng.choice([1, 2, 7, 20, 50, 100, 113]))
    S = 512.0
    cx, cy = rng.uniform(0, S, N), rng.uniform(0, S, N)
    w, h = rng.uniform(4, 60, N), rng.uniform(4, 60, N)
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    boxes = np.broadcast_to(boxes, (B, N, 4)).copy()
    sc = rng.uniform(0, 0.29, (B, N, C)).astype(np.float32)
    quant = float(rng.choice([16, 256, 1 << 16]))
    for b in range(B):
        n_hot = int(min(N, rng.choice([0, 3, max_ret, 2 * max_ret + 3, 8 * max_ret, 16 * max_ret + 50, N])))
        idx = rng.permutation(N)[:n_hot]
        share = rng.dirichlet(np.full(C, float(rng.choice([0.05, 0.5, 5.0]))))          # skewed or even class shares
        cls = rng.choice(C, n_hot, p=share)
        val = 0.3 + 0.7 * rng.random(n_hot)
        group = int(rng.choice([1, 1, 3, 12]))
        if group > 1 and n_hot >= group:                    # groups of identical boxes, one class, adjacent scores
            lead = idx[::group]
            for j in range(1, group):
                member = idx[j::group]
                m = len(member)
                boxes[b, member] = boxes[b, lead[:m]]
                cls[j::group] = cls[::group][:m]
                val[j::group] = val[::group][:m] - j * 2.0 ** -13
        val = np.floor(val * quant) / quant + 1 / (2 * quant)
        sc[b, idx, cls] = val.astype(np.float32)
    thr_i = float(rng.choice([0.3, 0.45, 0.7]))
    got = BB.nms(boxes, sc, iou_threshold=thr_i, max_ret=max_ret, return_keep=True)
    ref = c_oracle.decode_nms(None, boxes, sc, (S, S), do_clip=False, iou_threshold=thr_i, max_ret=max_ret, threads=4)
    b_, s_, l_, k_, n_ = (t.cpu().numpy() for t in got)
    np.testing.assert_array_equal(n_, ref[4])
    np.testing.assert_array_equal(k_, ref[3])
    np.testing.assert_array_equal(l_, ref[2])
    np.testing.assert_array_equal(s_, ref[1])
    np.testing.assert_array_equal(b_, ref[0])               # decoded boxes handed over: copied, not recomputed
    assert BB.set_nms_fused_tail(False) is True
    try:
        sep = BB.nms(boxes, sc, iou_threshold=thr_i, max_ret=max_ret, return_keep=True)
    finally:
        assert BB.set_nms_fused_tail(True) is False
    for x, y in zip(got, sep):
        assert torch.equal(x, y)


@pytest.mark.parametrize('seed', range(64))
def test_sweep_random_shapes_and_placements(seed):
    """The lockstep sweep (ODBOX_POLICY_STREAM; wide label rows, short GT lists, labels on a 16-byte boundary): random
    batch sizes up to 300 (valid rows beyond the LDS share), anchor counts that put image and row boundaries anywhere in
    the 4 KB pieces, padding before / between / behind valid rows, optional outputs, NaN-filled buffers with guard
    bands, against the C oracle."""
    rng = np.random.default_rng(9000 + seed)
    B = int(rng.choice([1, 2, 3, 5, 17, 64, 150, 300]))
    N = int(rng.choice([1, 2, 13, 63, 64, 65, 255, 257, 300, 777, 1025])) if B > 17 else int(rng.choice([1, 12, 13, 257, 1025, 4097, 9999]))
    G = int(rng.choice([1, 2, 3, 7, 20, 33, 64]))
    C = int(rng.choice([68, 69, 79, 80, 81, 90, 127, 255]))
    S = float(rng.choice([64, 100, 416]))
    cx, cy = rng.uniform(0, S, N), rng.uniform(0, S, N)
    w, h = rng.uniform(2, S / 2, N), rng.uniform(2, S / 2, N)
    anchors = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    gx, gy = rng.uniform(0, S, (B, G)), rng.uniform(0, S, (B, G))
    gw, gh = rng.uniform(1, S / 2, (B, G)), rng.uniform(1, S / 2, (B, G))
    gt = np.stack([gx - gw / 2, gy - gh / 2, gx + gw / 2, gy + gh / 2], -1).astype(np.float32)
    lab = rng.integers(-1, C + 2, (B, G)).astype(np.int32)
    for b in range(B):
        kind = rng.random()
        if kind < 0.5:
            gt[b, int(rng.integers(0, G + 1)):] = -1                 # valid prefix
        elif kind < 0.7:
            gt[b, rng.random(G) < 0.5] = -1                           # holes anywhere
        elif kind < 0.8:
            gt[b, :int(rng.integers(0, G + 1))] = -1                  # padding in front
        if G >= 2 and rng.random() < 0.4:
            gt[b, 1] = gt[b, 0]
        if rng.random() < 0.4:
            gt[b, 0] = anchors[int(rng.integers(0, N))]
        lab[b][gt[b, :, 2] < 0] = -1
    hw = (S, float(rng.choice([S, S - 16])))
    ref = c_oracle.match_encode(anchors, hw, gt, lab, C, threads=8)
    lib = _lib.load()
    dev = torch.device('cuda')
    want_reg, want_compact = rng.random() < 0.8, rng.random() < 0.5
    off_r, off_l = int(rng.integers(0, 40)), 4 * int(rng.integers(0, 10))
    guard = 64
    breg = torch.full((guard + off_r + B * N * 5 + guard,), float('nan'), device=dev)
    blab = torch.full((guard + off_l + B * N * (C + 1) + guard,), float('nan'), device=dev)
    reg = breg[guard + off_r: guard + off_r + B * N * 5]
    labels = blab[guard + off_l: guard + off_l + B * N * (C + 1)]
    idx = torch.full((B, N), -7, dtype=torch.int32, device=dev)
    state = torch.full((B, N), -7, dtype=torch.int8, device=dev)
    iou = torch.full((B, N), float('nan'), device=dev)
    a, g, l = (torch.as_tensor(x).to(dev) for x in (anchors, gt, lab))
    p = lambda t, on=True: ctypes.c_void_p(t.data_ptr()) if on else None      # noqa: E731
    prev = lib.odbox_set_output_policy(2)
    try:
        rc = lib.odbox_match_encode(p(a), p(g), p(l), B, N, G, C, hw[0], hw[1], 0.4, 0.5, p(reg, want_reg), p(labels),
                                    p(idx, want_compact), p(state, want_compact), p(iou, want_compact), None, _lib.stream_ptr())
        took = lib.odbox_match_encode_last_plan()
    finally:
        lib.odbox_set_output_policy(prev)
    assert rc == 0, lib.odbox_last_error()
    assert took == 2, 'not the sweep'
    torch.cuda.synchronize()
    np.testing.assert_array_equal(labels.view(B, N, C + 1).cpu().numpy(), ref['labels'])
    if want_reg:
        got = reg.view(B, N, 5).cpu().numpy()
        np.testing.assert_array_equal(got[..., 4], ref['regression'][..., 4])
        with np.errstate(invalid='ignore'):
            np.testing.assert_allclose(got[..., :4], ref['regression'][..., :4], rtol=1e-5, atol=0)
    else:
        assert torch.isnan(reg).all()
    if want_compact:
        np.testing.assert_array_equal(idx.cpu().numpy(), ref['match_idx'])
        np.testing.assert_array_equal(state.cpu().numpy(), ref['state'])
        np.testing.assert_array_equal(iou.cpu().numpy(), ref['max_iou'])
    else:
        assert (idx == -7).all() and (state == -7).all()
    for buf, off, n in ((breg, off_r, B * N * 5), (blab, off_l, B * N * (C + 1))):
        assert torch.isnan(buf[:guard + off]).all() and torch.isnan(buf[guard + off + n:]).all()

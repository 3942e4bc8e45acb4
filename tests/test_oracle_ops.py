"""Hand-computed known-answer tests for the restated torchvision ops (oracle/ops_ref.py).

These pin the one place where the oracle and the reference-under-standins share an author."""
import numpy as np
import torch

from oracle import ops_ref


def test_roi_align_constant_map_is_constant():
    x = torch.full((1, 3, 8, 10), 2.5)
    rois = torch.tensor([[0, 1.0, 1.0, 7.0, 6.0]])
    out = ops_ref.roi_align(x, rois, 4, 1.0, 2, False)
    assert out.shape == (1, 3, 4, 4)
    assert torch.allclose(out, torch.full_like(out, 2.5))


def test_roi_align_linear_ramp_legacy_pixel_model():
    # f(y, x) = x on a 1x1x4x8 map; roi [0,0,4,4] scale 1, 2x2 bins, sampling 2, aligned=False.
    # bin width 2, samples at x = 0.5, 1.5 (bin 0) and 2.5, 3.5 (bin 1); bilinear of a ramp is exact.
    x = torch.arange(8, dtype=torch.float32).view(1, 1, 1, 8).expand(1, 1, 4, 8).contiguous()
    rois = torch.tensor([[0, 0.0, 0.0, 4.0, 4.0]])
    out = ops_ref.roi_align(x, rois, 2, 1.0, 2, False)
    np.testing.assert_allclose(out[0, 0].numpy(), [[1.0, 3.0], [1.0, 3.0]], rtol=0, atol=1e-6)


def test_roi_align_min_size_one_and_scale():
    # zero-area roi at (8,8) with scale 0.25 -> start 2.0, roi w,h forced to 1 (aligned=False): one 1x1 bin,
    # samples at 2.25 and 2.75 in both axes -> mean of f = y*10+x at those points = 2.5*10 + 2.5 = 27.5
    yy, xx = torch.meshgrid(torch.arange(6.0), torch.arange(6.0), indexing="ij")
    x = (yy * 10 + xx).view(1, 1, 6, 6)
    out = ops_ref.roi_align(x, torch.tensor([[0, 8.0, 8.0, 8.0, 8.0]]), 1, 0.25, 2, False)
    np.testing.assert_allclose(out.item(), 27.5, atol=1e-5)


def test_roi_align_out_of_range_samples_are_zero_and_edges_clamp():
    x = torch.ones((1, 1, 4, 4))
    # roi far outside: every sample has x > W -> 0
    out = ops_ref.roi_align(x, torch.tensor([[0, 10.0, 0.0, 14.0, 4.0]]), 2, 1.0, 2, False)
    assert float(out.abs().max()) == 0.0
    # samples in (-1, 0) are clamped to 0 and samples in (H-1, H] collapse onto the last row -> still 1.0
    out = ops_ref.roi_align(x, torch.tensor([[0, -0.9, -0.9, 4.0, 4.0]]), 1, 1.0, 2, False)
    np.testing.assert_allclose(out.item(), 1.0, atol=1e-6)


def test_roi_align_batch_index():
    x = torch.stack([torch.zeros(1, 4, 4), torch.ones(1, 4, 4)])
    out = ops_ref.roi_align(x, torch.tensor([[1, 0.0, 0.0, 4.0, 4.0], [0, 0.0, 0.0, 4.0, 4.0]]), 2, 1.0, 2, False)
    assert float(out[0].min()) == 1.0 and float(out[1].max()) == 0.0


def test_nms_strict_threshold_and_order():
    # IoU(b0,b1) = 50/150 = 1/3 ; IoU(b0,b2) = 0.5 exactly (strict '>' keeps it at thr=0.5)
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0], [5.0, 0.0, 15.0, 10.0], [0.0, 0.0, 10.0, 5.0], [100.0, 100.0, 110.0, 110.0]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.95])
    keep = ops_ref.nms(boxes, scores, 0.5)
    assert keep.tolist() == [3, 0, 1, 2]
    keep = ops_ref.nms(boxes, scores, 0.3)
    assert keep.tolist() == [3, 0]  # b1 (1/3 > 0.3) and b2 (0.5 > 0.3) suppressed by b0


def test_nms_zero_area_boxes_nan_iou_not_suppressed():
    boxes = torch.tensor([[5.0, 5.0, 5.0, 5.0], [5.0, 5.0, 5.0, 5.0]])
    keep = ops_ref.nms(boxes, torch.tensor([0.5, 0.6]), 0.5)
    assert keep.tolist() == [1, 0]  # 0/0 = NaN > thr is False


def test_batched_nms_levels_do_not_interact_both_strategies():
    b = torch.tensor([[0.0, 0.0, 10.0, 10.0], [1.0, 1.0, 10.0, 10.0]])
    s = torch.tensor([0.9, 0.8])
    assert ops_ref.batched_nms(b, s, torch.tensor([0, 0]), 0.5).tolist() == [0]
    assert ops_ref.batched_nms(b, s, torch.tensor([0, 1]), 0.5).tolist() == [0, 1]
    # > 4000 elements -> per-class loop (CPU rule) ; result identical here
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 500, (1500, 2)).astype(np.float32)
    wh = rng.uniform(5, 80, (1500, 2)).astype(np.float32)
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], 1))
    scores = torch.from_numpy(rng.permutation(1500).astype(np.float32))
    lv = torch.from_numpy(rng.integers(0, 3, 1500))
    k1 = ops_ref.batched_nms(boxes, scores, lv, 0.5)
    ref = []
    for c in range(3):
        idx = torch.where(lv == c)[0]
        ref.append(idx[ops_ref.nms(boxes[idx], scores[idx], 0.5)])
    ref = torch.cat(ref)
    ref = ref[scores[ref].sort(descending=True)[1]]
    assert k1.tolist() == ref.tolist()


def test_c_backend_equals_numpy_backend_bit_for_bit():
    import pytest
    if ops_ref._C is None:
        pytest.skip("oracle/_ops_c.so not built")
    rng = np.random.default_rng(3)
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 16, 30, 44), generator=g)
    xy = rng.uniform(-20, 150, (200, 2)).astype(np.float32)
    wh = rng.uniform(0, 90, (200, 2)).astype(np.float32)
    rois = torch.from_numpy(np.concatenate([rng.integers(0, 2, (200, 1)).astype(np.float32), xy, xy + wh], 1))
    for P, sc in ((7, 0.25), (14, 0.125), (3, 1.0)):
        a = ops_ref.roi_align(x, rois, P, sc, 2, False, backend="c")
        b = ops_ref.roi_align(x, rois, P, sc, 2, False, backend="torch")
        assert torch.equal(a, b)
    boxes = torch.from_numpy(np.concatenate([xy, xy + wh], 1))
    scores = torch.from_numpy(rng.random(200).astype(np.float32))
    for thr in (0.3, 0.5, 0.7):
        assert torch.equal(ops_ref.nms(boxes, scores, thr, backend="c"), ops_ref.nms(boxes, scores, thr, backend="numpy"))


def test_bilinear_restatement_is_bit_exact():
    """The arithmetic dp_iuv_extract implements for the visualiser's F.interpolate(bilinear, align_corners=False)
    (visualizer.py:14-16,24-25): src = fma(scale, dst + 0.5, -0.5), value = fma(wy0, fma(wx0, a, wx1*b), wy1 * fma(wx0, c, wx1*d)).
    numpy float64 products of float32 operands are exact, so fma() below rounds once like the hardware instruction;
    the result must equal torch's CPU kernel bit for bit (that is what the -m gpu tests then hold the HIP kernel to)."""
    import torch
    import torch.nn.functional as F
    f32 = np.float32

    def fma(p, q, r):
        return (np.asarray(p, dtype=np.float64) * np.asarray(q, dtype=np.float64) + np.asarray(r, dtype=np.float64)).astype(np.float32)

    def axis(n_out, n_in):
        scale = f32(n_in) / f32(n_out)
        src = np.maximum(fma(scale, (np.arange(n_out, dtype=np.float32) + f32(0.5)).astype(np.float32), f32(-0.5)), f32(0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        lam = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
        return i0, np.minimum(i0 + 1, n_in - 1), (f32(1) - lam).astype(np.float32), lam

    x = torch.randn((1, 5, 112, 112), generator=torch.Generator().manual_seed(0))
    X = x.numpy()[0]
    for h, w in ((57, 83), (200, 131), (13, 300), (112, 112), (1, 1), (300, 7)):
        ref = F.interpolate(x, (h, w), mode="bilinear", align_corners=False).numpy()[0]
        y0, y1, wy0, wy1 = axis(h, 112)
        x0, x1, wx0, wx1 = axis(w, 112)
        a, b, c, d = X[:, y0][:, :, x0], X[:, y0][:, :, x1], X[:, y1][:, :, x0], X[:, y1][:, :, x1]
        WX0, WX1, WY0, WY1 = wx0[None, None, :], wx1[None, None, :], wy0[None, :, None], wy1[None, :, None]
        t0 = fma(WX0, a, (WX1 * b).astype(np.float32))
        t1 = fma(WX0, c, (WX1 * d).astype(np.float32))
        got = fma(WY0, t0, (WY1 * t1).astype(np.float32))
        assert np.array_equal(got, ref), (h, w, float(np.abs(got - ref).max()))

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


# ---- round 6: more known answers derived by hand from torchvision 0.16.2's published kernels (csrc/ops/cpu/roi_align_kernel.cpp,
# roi_align_common.h: pre_calc_for_bilinear_interpolate; csrc/ops/cpu/nms_kernel.cpp; ops/boxes.py: batched_nms). f(y, x) = 10 y + x + 1.
def _ramp(h=4, w=4):
    yy, xx = torch.meshgrid(torch.arange(float(h)), torch.arange(float(w)), indexing="ij")
    return (10 * yy + xx + 1).view(1, 1, h, w)


def test_roi_align_sample_exactly_at_minus_one_is_inside():
    # `if (y < -1.0 || y > height)` is strict: a sample at y = -1.0 is valid and clamps to row 0. One 1x1 bin, sampling 1:
    # roi y in [-1.5, -0.5] -> bin height 1 -> y = -1.5 + 0.5 = -1.0 ; x in [0.5, 1.5] -> x = 1.0 -> f(0, 1) = 2
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 0.5, -1.5, 1.5, -0.5]]), 1, 1.0, 1, False)
    assert out.item() == 2.0
    # ... and a quarter further out (y = -1.25 < -1) contributes nothing
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 0.5, -1.75, 1.5, -0.75]]), 1, 1.0, 1, False)
    assert out.item() == 0.0


def test_roi_align_sample_exactly_at_height_and_width_is_inside():
    # y = H = 4.0 is not `> height`: y_low = 4 >= H - 1 collapses onto the last row (y_low = y_high = 3, y = 3): f(3, 1) = 32
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 0.5, 3.5, 1.5, 4.5]]), 1, 1.0, 1, False)
    assert out.item() == 32.0
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 0.5, 3.75, 1.5, 4.75]]), 1, 1.0, 1, False)      # y = 4.25 > H
    assert out.item() == 0.0
    # the same along x: x = W = 4.0 -> column 3; y = 1.0 -> f(1, 3) = 14
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 3.5, 0.5, 4.5, 1.5]]), 1, 1.0, 1, False)
    assert out.item() == 14.0


def test_roi_align_last_row_collapse_drops_the_fraction():
    # y = 3.4: y_low = 3 >= H - 1 -> y_high = y_low = 3 AND y = 3 (ly = 0): no interpolation towards a row that does not exist.
    # x = 1.5 interpolates: 0.5 f(3, 1) + 0.5 f(3, 2) = 0.5 * 32 + 0.5 * 33 = 32.5
    out = ops_ref.roi_align(_ramp(), torch.tensor([[0, 1.0, 2.9, 2.0, 3.9]]), 1, 1.0, 1, False)
    np.testing.assert_allclose(out.item(), 32.5, rtol=0, atol=1e-6)


def test_roi_align_count_is_the_grid_size_even_when_samples_fall_outside():
    # one bin over x in [3, 7] on a 4-wide map of ones, sampling 2: x samples 4.0 (inside: == W, clamps to column 3) and 6.0 (outside);
    # y samples 0.5 and 1.5 (inside). Sum = 2 valid samples x 1.0, divided by count = 2 x 2 = 4 -> 0.5
    out = ops_ref.roi_align(torch.ones((1, 1, 4, 4)), torch.tensor([[0, 3.0, 0.0, 7.0, 2.0]]), 1, 1.0, 2, False)
    assert out.item() == 0.5


def test_roi_align_aligned_flag_shifts_by_half_a_pixel_and_keeps_small_rois():
    # aligned = True: start = 2 - 0.5 = 1.5, width 2 (no minimum of 1), sampling 2: x samples 2.0 and 3.0 on f = x ramp -> 2.5
    x = torch.arange(8, dtype=torch.float32).view(1, 1, 1, 8).expand(1, 1, 4, 8).contiguous()
    out = ops_ref.roi_align(x, torch.tensor([[0, 2.0, 1.0, 4.0, 3.0]]), 1, 1.0, 2, True)
    np.testing.assert_allclose(out.item(), 2.5, rtol=0, atol=1e-6)
    # ... and a zero-width roi stays zero-width there (both samples at 1.5: 1.5), where aligned = False forces width 1 (samples 2.25, 2.75: 2.5)
    out = ops_ref.roi_align(x, torch.tensor([[0, 2.0, 1.0, 2.0, 3.0]]), 1, 1.0, 2, True)
    np.testing.assert_allclose(out.item(), 1.5, rtol=0, atol=1e-6)
    out = ops_ref.roi_align(x, torch.tensor([[0, 2.0, 1.0, 2.0, 3.0]]), 1, 1.0, 2, False)
    np.testing.assert_allclose(out.item(), 2.5, rtol=0, atol=1e-6)


def test_roi_align_sample_order_inside_a_bin_is_row_major():
    # fp32 sums are order dependent: values 1e8, 1, -1e8, 1 at the four samples of one bin (iy, ix) = (0,0), (0,1), (1,0), (1,1) give
    # ((1e8 + 1) - 1e8) + 1 = 1 in row-major order (1e8 + 1 rounds to 1e8), /4 = 0.25 - any other order gives 0.5 or 0
    x4 = torch.zeros((1, 1, 4, 4))
    x4[0, 0, 0, 0], x4[0, 0, 0, 2], x4[0, 0, 2, 0], x4[0, 0, 2, 2] = 1e8, 1.0, -1e8, 1.0
    out = ops_ref.roi_align(x4, torch.tensor([[0, -1.0, -1.0, 3.0, 3.0]]), 1, 1.0, 2, False)      # samples at 0.0 and 2.0 in both axes
    assert out.item() == 0.25


def test_nms_threshold_is_compared_in_double_like_torchvision():
    # inter 50, union 150: the float IoU is float32(1/3) = 0.3333333432674408 > 1/3 (double): torchvision suppresses; a threshold naively cast to
    # float would be that very float and keep the box
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0], [5.0, 0.0, 15.0, 10.0]])
    scores = torch.tensor([0.9, 0.8])
    assert float(np.float32(50.0) / np.float32(150.0)) > 1.0 / 3.0
    assert ops_ref.nms(boxes, scores, 1.0 / 3.0, backend="numpy").tolist() == [0]
    if ops_ref._C is not None:
        assert ops_ref.nms(boxes, scores, 1.0 / 3.0, backend="c").tolist() == [0]
    # thresholds of the model: 0.7 rounds DOWN as a float (0.699999988) and 0.5 is exact - unchanged by the rule
    assert float(ops_ref.nms_threshold_f32(0.7)) == float(np.float32(0.7)) and float(ops_ref.nms_threshold_f32(0.5)) == 0.5
    assert float(ops_ref.nms_threshold_f32(0.3)) < 0.3 < float(np.float32(0.3))


def test_nms_iou_exactly_at_the_model_thresholds_is_kept():
    # IoU = 7 / 10 exactly in float: boxes [0,0,10,10] and [0,0,10,7]: inter 70, union 100 -> 0.7f = 0.699999988 ; 0.699999988 > 0.7 is false
    boxes = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 7.0]])
    assert ops_ref.nms(boxes, torch.tensor([0.9, 0.8]), 0.7).tolist() == [0, 1]
    # one ulp more overlap is suppressed: [0,0,10,7.000001] -> inter 70.00001, IoU 0.7000001 > 0.7
    boxes2 = torch.tensor([[0.0, 0.0, 10.0, 10.0], [0.0, 0.0, 10.0, 7.00001]])
    assert ops_ref.nms(boxes2, torch.tensor([0.9, 0.8]), 0.7).tolist() == [0]


def test_nms_equal_scores_keep_input_order_and_zero_area_never_suppresses():
    # stable sort: equal scores are visited in index order; a zero-area box inside a kept one has IoU 0 / area = 0 (kept), two identical
    # zero-area boxes have 0 / 0 = NaN (kept: NaN > thr is false)
    boxes = torch.tensor([[0.0, 0.0, 4.0, 4.0], [0.0, 0.0, 4.0, 4.0], [1.0, 1.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0]])
    scores = torch.tensor([0.5, 0.5, 0.5, 0.5])
    assert ops_ref.nms(boxes, scores, 0.5).tolist() == [0, 2, 3]


def test_batched_nms_strategy_switch_at_4000_elements_with_two_groups():
    # ops/boxes.py: `if boxes.numel() > 4000` (CPU) runs one nms per group, else ONE nms on boxes + idx * (max + 1). 1000 boxes = 4000
    # elements is still the trick; 1002 boxes the loop. Two groups with IDENTICAL coordinates: under either strategy a box only ever
    # meets boxes of its own group.
    n = 500
    base = torch.zeros((n, 4))
    base[:, 0] = torch.arange(n) * 10.0
    base[:, 2] = base[:, 0] + 4.0
    base[:, 3] = 4.0
    boxes = torch.cat([base, base])                       # group 0 and group 1: identical coordinates
    idxs = torch.cat([torch.zeros(n), torch.ones(n)]).long()
    scores = torch.cat([torch.linspace(1.0, 0.6, n), torch.linspace(0.59, 0.2, n)])
    assert boxes.numel() == 4000
    k_trick = ops_ref.batched_nms(boxes, scores, idxs, 0.5)
    assert k_trick.tolist() == list(range(2 * n))         # nothing overlaps inside a group, groups never interact
    # one more pair of coinciding boxes (1002 boxes = 4008 elements) switches to the loop: same answer
    extra = torch.tensor([[0.0, 0.0, 4.0, 4.0]])
    boxes2 = torch.cat([boxes, extra, extra])
    idxs2 = torch.cat([idxs, torch.tensor([0, 1])])
    scores2 = torch.cat([scores, torch.tensor([0.1, 0.05])])
    k_loop = ops_ref.batched_nms(boxes2, scores2, idxs2, 0.5)
    # the two extra boxes coincide with box 0 of their OWN group (IoU 1 > 0.5): suppressed in both groups, everything else kept in score order
    assert k_loop.tolist() == list(range(2 * n))
    # and with the groups swapped for the extras each one only meets the other group's box - still suppressed by its own group's twin
    idxs3 = torch.cat([idxs, torch.tensor([1, 0])])
    assert ops_ref.batched_nms(boxes2, scores2, idxs3, 0.5).tolist() == list(range(2 * n))


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

"""ORACLE (test infrastructure, never imported by the product package).

CPU restatement of the two torchvision ops the reference's hot path calls:

  * ``roi_align``   - called at /root/reference/detectron2/layers/roi_align.py:58-65
  * ``nms`` / ``batched_nms`` - called at /root/reference/detectron2/layers/nms.py:20

torchvision (pinned ``torchvision~=0.16.2`` in /root/reference/requirements.txt:2) is a
third-party dependency that is NOT vendored under /root/reference and NOT installed in
this image, so its published CPU algorithms are restated here:

  * torchvision/csrc/ops/cpu/roi_align_kernel.cpp + roi_align_common.h
    (``pre_calc_for_bilinear_interpolate``): legacy pixel model when ``aligned=False``
    (roi w,h clamped to >= 1), ``sampling_ratio`` g x g samples per bin, sample outside
    [-1, H] x [-1, W] contributes 0, coordinates clamped to >= 0, low index = (int)coord,
    at the last row/col both indices collapse to H-1 / W-1; output = sum / (g*g).
  * torchvision/csrc/ops/cpu/nms_kernel.cpp: stable descending sort by score, greedy,
    suppress when inter / (area_i + area_j - inter) > thr (strict), areas and IoU in fp32.
  * torchvision/ops/boxes.py ``batched_nms``: coordinate-offset trick when
    ``boxes.numel() <= 4000`` on CPU (20000 on GPU), otherwise the per-class loop
    (``_batched_nms_vanilla``) followed by a descending score sort of the kept indices.

Because the oracle and the reference share these two restatements (the reference is run
here with them as its torchvision stand-in), they are additionally pinned by
hand-computed known-answer tests in tests/test_oracle_ops.py.
"""
import ctypes
import os

import numpy as np
import torch

# Optional C build of the same two kernels (oracle/ops_c.c -> oracle/_ops_c.so, `make -C oracle`): identical
# arithmetic and operation order, only faster; tests/test_oracle_ops.py checks C == numpy/torch bit for bit.
_C = None
_so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ops_c.so")
if os.path.exists(_so) and os.environ.get("ORACLE_NO_C", "0") != "1":
    try:
        _C = ctypes.CDLL(_so)
        _C.oracle_nms_f32.restype = ctypes.c_int
    except OSError:
        _C = None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False, backend=None):
    use_c = (_C is not None) if backend is None else (backend == "c")
    if use_c and input.dtype == torch.float32 and rois.shape[0] > 0:
        if isinstance(output_size, (tuple, list)):
            assert output_size[0] == output_size[1]
            output_size = output_size[0]
        x = input.contiguous()
        r = rois.to(torch.float32).contiguous()
        N, Cc, H, W = x.shape
        K = r.shape[0]
        out = torch.empty((K, Cc, int(output_size), int(output_size)), dtype=torch.float32)
        _C.oracle_roi_align_f32(ctypes.c_void_p(x.data_ptr()), N, Cc, H, W, ctypes.c_void_p(r.data_ptr()), K, int(output_size),
                                ctypes.c_float(spatial_scale), int(sampling_ratio), int(bool(aligned)), ctypes.c_void_p(out.data_ptr()))
        return out
    return _roi_align_torch(input, rois, output_size, spatial_scale, sampling_ratio, aligned)


def nms_threshold_f32(iou_threshold):
    """torchvision's CPU kernel (torchvision 0.16.2 csrc/ops/cpu/nms_kernel.cpp: `auto ovr = inter / (...); if (ovr > iou_threshold)`) compares
    the FLOAT IoU with the DOUBLE threshold. For a float x:  x > t_double  <=>  x > the largest float not above t_double - so the whole
    comparison can stay in float when the threshold is rounded DOWN (0.7 -> 0.699999988, 0.5 exact; a plain float(0.3) = 0.300000012 would keep
    a pair whose IoU is exactly that float where torchvision suppresses it)."""
    t = np.float32(iou_threshold)
    if float(t) > float(iou_threshold):
        t = np.nextafter(t, np.float32(-np.inf))
    return np.float32(t)


def nms(boxes, scores, iou_threshold, backend=None):
    use_c = (_C is not None) if backend is None else (backend == "c")
    if use_c and boxes.shape[0] > 0:
        b = boxes.detach().to(torch.float32).contiguous()
        order = torch.sort(scores.detach().to(torch.float32), descending=True, stable=True)[1].contiguous()
        keep = torch.empty((b.shape[0],), dtype=torch.int64)
        nk = _C.oracle_nms_f32(ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(order.data_ptr()), int(b.shape[0]),
                               ctypes.c_float(float(nms_threshold_f32(iou_threshold))), ctypes.c_void_p(keep.data_ptr()))
        return keep[:nk].clone()
    return _nms_numpy(boxes, scores, iou_threshold)


def _roi_align_torch(input, rois, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    """input [N,C,H,W] float, rois [K,5] (batch, x1, y1, x2, y2) -> [K,C,ph,pw]."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    ph, pw = int(output_size[0]), int(output_size[1])
    N, C, H, W = input.shape
    K = rois.shape[0]
    dt = input.dtype
    out = torch.zeros((K, C, ph, pw), dtype=dt)
    if K == 0:
        return out
    assert sampling_ratio > 0, "hot path always uses sampling_ratio=2"
    g = int(sampling_ratio)
    rois = rois.to(dt)
    bidx = rois[:, 0].to(torch.int64)
    off = 0.5 if aligned else 0.0
    scale = torch.tensor(spatial_scale, dtype=dt)
    x1 = rois[:, 1] * scale - off
    y1 = rois[:, 2] * scale - off
    x2 = rois[:, 3] * scale - off
    y2 = rois[:, 4] * scale - off
    rw = x2 - x1
    rh = y2 - y1
    if not aligned:
        rw = torch.clamp(rw, min=1.0)
        rh = torch.clamp(rh, min=1.0)
    bin_h = rh / ph
    bin_w = rw / pw
    # sample coordinates: [K, ph, g] and [K, pw, g]
    iy = torch.arange(g, dtype=dt) + 0.5
    py = torch.arange(ph, dtype=dt)
    px = torch.arange(pw, dtype=dt)
    # yy = roi_start_h + ph * bin_size_h + (iy + .5) * bin_size_h / grid   (same op order as the kernel)
    yy = y1[:, None, None] + py[None, :, None] * bin_h[:, None, None] + iy[None, None, :] * bin_h[:, None, None] / g
    xx = x1[:, None, None] + px[None, :, None] * bin_w[:, None, None] + iy[None, None, :] * bin_w[:, None, None] / g
    yy = yy.reshape(K, ph * g)
    xx = xx.reshape(K, pw * g)

    def prep(c, size):
        invalid = (c < -1.0) | (c > size)
        c = torch.clamp(c, min=0.0)
        lo = c.to(torch.int64)  # (int) truncation of a non-negative value
        last = lo >= size - 1
        lo = torch.where(last, torch.full_like(lo, size - 1), lo)
        hi = torch.where(last, lo, lo + 1)
        c = torch.where(last, lo.to(dt), c)
        l = c - lo.to(dt)
        h = 1.0 - l
        return invalid, lo, hi, l, h

    inv_y, ylo, yhi, ly, hy = prep(yy, H)
    inv_x, xlo, xhi, lx, hx = prep(xx, W)
    count = float(max(g * g, 1))
    # process per ROI chunk to bound memory
    for k0 in range(0, K, 64):
        k1 = min(K, k0 + 64)
        kk = k1 - k0
        feat = input[bidx[k0:k1]]  # [kk,C,H,W]
        Yl = ylo[k0:k1][:, :, None].expand(kk, ph * g, pw * g)
        Yh = yhi[k0:k1][:, :, None].expand(kk, ph * g, pw * g)
        Xl = xlo[k0:k1][:, None, :].expand(kk, ph * g, pw * g)
        Xh = xhi[k0:k1][:, None, :].expand(kk, ph * g, pw * g)
        flat = feat.reshape(kk, C, H * W)

        def gather(Y, X):
            idx = (Y * W + X).reshape(kk, 1, -1).expand(kk, C, -1)
            return torch.gather(flat, 2, idx).reshape(kk, C, ph * g, pw * g)

        w1 = (hy[k0:k1][:, :, None] * hx[k0:k1][:, None, :])[:, None]
        w2 = (hy[k0:k1][:, :, None] * lx[k0:k1][:, None, :])[:, None]
        w3 = (ly[k0:k1][:, :, None] * hx[k0:k1][:, None, :])[:, None]
        w4 = (ly[k0:k1][:, :, None] * lx[k0:k1][:, None, :])[:, None]
        val = w1 * gather(Yl, Xl) + w2 * gather(Yl, Xh) + w3 * gather(Yh, Xl) + w4 * gather(Yh, Xh)
        invalid = (inv_y[k0:k1][:, :, None] | inv_x[k0:k1][:, None, :])[:, None]
        val = torch.where(invalid, torch.zeros((), dtype=dt), val)
        # accumulate the g x g samples of each bin in (iy, ix) order like the kernel does
        val = val.reshape(kk, C, ph, g, pw, g)
        acc = torch.zeros((kk, C, ph, pw), dtype=dt)
        for a in range(g):
            for b in range(g):
                acc = acc + val[:, :, :, a, :, b]
        out[k0:k1] = acc / count
    return out


def _nms_numpy(boxes, scores, iou_threshold):
    """Greedy NMS; returns kept indices (int64) in descending score order."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    b = boxes.detach().to(torch.float32).numpy()
    s = scores.detach().to(torch.float32)
    order = torch.sort(s, descending=True, stable=True)[1].numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    thr = nms_threshold_f32(iou_threshold)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = np.float32(0)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        if rest.size == 0:
            break
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(zero, xx2 - xx1)
        h = np.maximum(zero, yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def _uses_coordinate_trick(boxes, trick_max_numel):
    # torchvision/ops/boxes.py: vanilla loop iff numel > 4000 on CPU (20000 on GPU)
    return not (boxes.numel() > trick_max_numel)


def batched_nms(boxes, scores, idxs, iou_threshold, trick_max_numel=4000):
    """trick_max_numel: torchvision's strategy switch - 4000 where the reference runs on the CPU (the goldens), 20000 for its
    CUDA mode (run.py:22-29)."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if _uses_coordinate_trick(boxes, trick_max_numel):
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
        return nms(boxes_for_nms, scores, iou_threshold)
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr = torch.where(idxs == class_id)[0]
        k = nms(boxes[curr], scores[curr], iou_threshold)
        keep_mask[curr[k]] = True
    keep_indices = torch.where(keep_mask)[0]
    return keep_indices[scores[keep_indices].sort(descending=True)[1]]

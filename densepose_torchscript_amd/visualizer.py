"""GPU counterpart of the reference's ``DensePoseResultExtractor`` (/root/reference/visualizer.py:10-56).

For every detection: bilinear resample (align_corners=False) of the coarse / fine / U / V maps to the integer-truncated
box size, ``labels = argmax(fine) * (argmax(coarse) > 0)`` and the U, V of the winning part - one ``dp_iuv_extract``
launch for all detections of a frame, instead of R x 4 ``F.interpolate`` calls on (R, C, S, S) tensors. The D2H volume
drops from R x 77 x S x S floats to 9 bytes per box pixel. Drawing (cv2 colour maps) stays out of scope.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def boxes_xywh(outputs):
    """visualizer.py:40-43,34: xyxy -> xywh, truncated to int64, w/h at least 1."""
    b = outputs["pred_boxes"].detach().float().cpu().clone()
    b[:, 2:] -= b[:, :2]
    b = b.long()
    b[:, 2:] = b[:, 2:].clamp(min=1)
    return b


def extract_iuv(outputs, stream=None):
    """outputs: the predictor's dict (tensors on the GPU). Returns (results, boxes_xywh) with
    results[i] = {"labels": uint8 [h, w], "uv": float32 [2, h, w]} on the GPU, like predictor_output_to_result."""
    lib = L.load()
    coarse, fine = outputs["pred_densepose_coarse_segm"], outputs["pred_densepose_fine_segm"]
    u, v = outputs["pred_densepose_u"], outputs["pred_densepose_v"]
    R = int(coarse.shape[0])
    xywh = boxes_xywh(outputs)
    if R == 0:
        return [], xywh
    dev = coarse.device
    for t in (coarse, fine, u, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    hw = (xywh[:, 2] * xywh[:, 3]).numpy()
    offs = np.zeros((R,), dtype=np.int64)
    offs[1:] = np.cumsum(hw)[:-1]
    total = int(hw.sum())
    labels = torch.empty((total,), dtype=torch.uint8, device=dev)
    uv = torch.empty((2 * total,), dtype=torch.float32, device=dev)
    box_d = xywh.to(torch.int32).to(dev)
    off_d = torch.from_numpy(offs).to(dev)
    p = L.IuvExtractParams()
    p.coarse, p.fine, p.u, p.v = coarse.data_ptr(), fine.data_ptr(), u.data_ptr(), v.data_ptr()
    p.R, p.S, p.n_coarse, p.n_fine = R, int(coarse.shape[-1]), int(coarse.shape[1]), int(fine.shape[1])
    p.box_xywh, p.out_offset, p.labels, p.uv, p.max_hw = box_d.data_ptr(), off_d.data_ptr(), labels.data_ptr(), uv.data_ptr(), int(hw.max())
    s = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
    L.check(lib.dp_iuv_extract(C.byref(p), C.c_void_p(s)), "dp_iuv_extract")
    results = []
    for r in range(R):
        w, h = int(xywh[r, 2]), int(xywh[r, 3])
        o = int(offs[r])
        results.append({"labels": labels[o:o + h * w].view(h, w), "uv": uv[2 * o:2 * o + 2 * h * w].view(2, h, w)})
    return results, xywh


def iuv_image(results, xywh, height, width):
    """Compose the per-detection IUV crops into one uint8 [3, H, W] array (I, U*255, V*255) in frame coordinates -
    the array the reference's visualiser colour-maps (visualizer.py:124-131), without the drawing."""
    out = np.zeros((3, height, width), dtype=np.uint8)
    for r, res in enumerate(results):
        x, y, w, h = [int(t) for t in xywh[r]]
        lab = res["labels"].cpu().numpy()
        uvv = (res["uv"].cpu().numpy() * 255.0).clip(0, 255).astype(np.uint8)
        x1, y1 = min(x + w, width), min(y + h, height)
        if x1 <= x or y1 <= y:
            continue
        crop = np.concatenate([lab[None], uvv], axis=0)[:, : y1 - y, : x1 - x]
        m = crop[0] > 0
        for c in range(3):
            out[c, y:y1, x:x1][m] = crop[c][m]
    return out

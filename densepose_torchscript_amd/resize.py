"""GPU-side shortest-edge resize, bit-exact with the CPU uint8 kernel the reference runs at
/root/reference/detectron2/engine/defaults.py:89 (``F.interpolate(uint8, scale_factor=k, bilinear)``).

ATen resizes uint8 images with a two-pass (horizontal, then vertical) fixed-point filter; the weight tables
are tiny (O(H + W)) and are computed here on the host in float64 exactly as SURVEY App. E measured them
(validated against torch on 18 shape/scale combinations: 0 mismatching bytes); the per-pixel work is the
``dp_resize_u8_bilinear`` HIP kernel.
"""
import ctypes as C
from functools import lru_cache

import numpy as np
import torch

from . import lib as L


@lru_cache(maxsize=64)
def axis_table(n_in, n_out, k):
    """-> (int32 [n_out, 4] = {i0, i1, W0, W1}, precision bits)."""
    o = np.arange(n_out, dtype=np.float64)
    src = np.maximum((1.0 / k) * (o + 0.5) - 0.5, 0.0)
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    lam = src - i0
    w0, w1 = 1.0 - lam, lam
    wmax = max(float(w0.max()), float(w1.max()))
    p = 0
    while p < 22 and int(0.5 + wmax * 2 ** (p + 1)) < 2 ** 15:
        p += 1
    tab = np.stack([i0, i1, np.floor(0.5 + w0 * 2 ** p).astype(np.int64), np.floor(0.5 + w1 * 2 ** p).astype(np.int64)], axis=1)
    return tab.astype(np.int32), p


def output_size(h, w, k):
    return int(np.floor(h * k)), int(np.floor(w * k))  # Q5: floor, scale_factor used as given


def resize_u8_host_model(img_chw, k):
    """numpy model of the two passes (tests / documentation of the algorithm)."""
    _, H, W = img_chw.shape
    oh, ow = output_size(H, W, k)
    xt, px = axis_table(W, ow, k)
    t = (xt[:, 2] * img_chw[:, :, xt[:, 0]].astype(np.int64) + xt[:, 3] * img_chw[:, :, xt[:, 1]].astype(np.int64) + (1 << (px - 1))) >> px
    t = np.clip(t, 0, 255)
    yt, py = axis_table(H, oh, k)
    r = (yt[None, :, 2, None] * t[:, yt[:, 0], :] + yt[None, :, 3, None] * t[:, yt[:, 1], :] + (1 << (py - 1))) >> py
    return np.clip(r, 0, 255).astype(np.uint8)


def resize_u8_device_batch(engine, frames, k, src_hwc=False):
    """frames: list of uint8 device tensors of ONE geometry, each [3,H,W] (or [H,W,3] with src_hwc).
    Returns uint8 [n,3,oh,ow] on the device: one launch per pass for the whole list, written straight into the batch."""
    frames = [f.contiguous() for f in frames]
    f0 = frames[0]
    if src_hwc:
        H, W = int(f0.shape[0]), int(f0.shape[1])
    else:
        H, W = int(f0.shape[1]), int(f0.shape[2])
    assert all(f.shape == f0.shape and f.dtype == torch.uint8 and f.is_cuda for f in frames)
    n = len(frames)
    oh, ow = output_size(H, W, k)
    xt, px = axis_table(W, ow, k)
    yt, py = axis_table(H, oh, k)
    dev = engine.device
    cache = engine.__dict__.setdefault("_resize_tables", {})
    key = (H, W, oh, ow, k)
    if key not in cache:  # tables are uploaded once per frame geometry
        cache[key] = (torch.from_numpy(xt).to(dev), torch.from_numpy(yt).to(dev))
    xtab, ytab = cache[key]
    tmp = torch.empty((n, 3, H, ow), dtype=torch.uint8, device=dev)
    dst = torch.empty((n, 3, oh, ow), dtype=torch.uint8, device=dev)
    p = L.ResizeParams()
    p.src, p.tmp, p.dst = None, tmp.data_ptr(), dst.data_ptr()
    p.H, p.W, p.oh, p.ow, p.src_hwc = H, W, oh, ow, 1 if src_hwc else 0
    p.xtab, p.ytab, p.xprec, p.yprec = xtab.data_ptr(), ytab.data_ptr(), px, py
    srcs = (C.c_void_p * n)(*[f.data_ptr() for f in frames])
    L.check(engine.lib.dp_resize_u8_bilinear_batch(C.byref(p), srcs, n, engine._stream()), "dp_resize_u8_bilinear_batch")
    return dst


def resize_u8_device(engine, img, k, src_hwc=False):
    """img: uint8 device tensor [3,H,W] (or [H,W,3] with src_hwc). Returns uint8 [3,oh,ow] on the device."""
    return resize_u8_device_batch(engine, [img], k, src_hwc)[0]


class FusedResize:
    """Frames of ONE geometry whose scale is not 1, to be resized AND preprocessed by dp_resize_preprocess_u8_batch: the engine calls
    run(x) with the paired-layout tensor its stem reads ([n, Hp, Wp / 2 + 3, 8], engine.preprocess) - the horizontal pass, then one
    launch for vertical pass + normalise + pad + layout; the resized uint8 batch is never materialised."""

    def __init__(self, engine, frames, k, src_hwc):
        self.frames = [f.contiguous() for f in frames]
        f0 = self.frames[0]
        self.src_hwc = bool(src_hwc)
        self.H, self.W = (int(f0.shape[0]), int(f0.shape[1])) if src_hwc else (int(f0.shape[1]), int(f0.shape[2]))
        assert all(f.shape == f0.shape and f.dtype == torch.uint8 and f.is_cuda for f in self.frames) and len(self.frames) <= 64
        self.k = k
        self.oh, self.ow = output_size(self.H, self.W, k)
        self.n = len(self.frames)
        self.shape = (self.n, 3, self.oh, self.ow)       # what the resized uint8 batch would be

    def run(self, engine, x):
        xt, px = axis_table(self.W, self.ow, self.k)
        yt, py = axis_table(self.H, self.oh, self.k)
        dev = engine.device
        cache = engine.__dict__.setdefault("_resize_tables", {})
        key = (self.H, self.W, self.oh, self.ow, self.k)
        if key not in cache:
            cache[key] = (torch.from_numpy(xt).to(dev), torch.from_numpy(yt).to(dev))
        xtab, ytab = cache[key]
        tmp = torch.empty((self.n, 3, self.H, self.ow), dtype=torch.uint8, device=dev)
        p = L.ResizeParams()
        p.src, p.tmp, p.dst = None, tmp.data_ptr(), None
        p.H, p.W, p.oh, p.ow, p.src_hwc = self.H, self.W, self.oh, self.ow, 1 if self.src_hwc else 0
        p.xtab, p.ytab, p.xprec, p.yprec = xtab.data_ptr(), ytab.data_ptr(), px, py
        q = L.PreprocessParams()
        Hp, Wq = int(x.shape[1]), int(x.shape[2])
        q.src, q.dst, q.paired, q.src_hwc = None, x.data_ptr(), 1, 0
        q.n_img, q.h, q.w, q.Hp, q.Wp, q.dtype = self.n, self.oh, self.ow, Hp, 2 * (Wq - 3), engine.dt
        for i in range(3):
            q.mean[i] = engine.cfg.pixel_mean[i]
            q.std[i] = engine.cfg.pixel_std[i]
        srcs = (C.c_void_p * self.n)(*[f.data_ptr() for f in self.frames])
        L.check(engine.lib.dp_resize_preprocess_u8_batch(C.byref(p), srcs, self.n, C.byref(q), engine._stream()), "dp_resize_preprocess_u8_batch")

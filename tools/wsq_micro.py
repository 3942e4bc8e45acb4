"""The 256 -> 256 3x3 layers on conv3x3_wsq_kernel (dp_conv_wq.hip, kernel class 10: v_mfma_f32_32x32x16, one wave per SIMD) against
conv3x3_wsr_kernel<256> (class 6) and the LDS-ring kernel: correctness (torch fp64 on small maps, the older kernels within two
rounding steps on the large ones, image alone == image in the batch) and timing (warm clocks, median of passes).
usage: python tools/wsq_micro.py [batch] [dtype]     CHECK=0 skips the correctness part, SHAPES="200x336,50x84" picks levels"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch
import torch.nn.functional as F

from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd import lib as L
from densepose_torchscript_amd.engine import Act, Engine
from densepose_torchscript_amd.pack import conv_from_oihw

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype=dt)
g = torch.Generator().manual_seed(0)
Cc = 256
w = (torch.randn((Cc, Cc, 3, 3), generator=g) * (2.0 / (Cc * 9)) ** 0.5).to(tdt).float()
b = torch.randn((Cc,), generator=g) * 0.2
layer = conv_from_oihw("micro", w.numpy(), b.numpy(), Cc, 1, 1, 1, e.dt, e.device)
ulp = 2.0 ** -8 if dt == "bf16" else 2.0 ** -11


def klass(x):
    p = L.ConvParams()
    p.N, p.H, p.W, p.Cin, p.Ho, p.Wo, p.Cout, p.Cout_w, p.Kpad = x.N, x.H, x.W, Cc, x.H, x.W, Cc, Cc, 9 * Cc
    p.stride, p.ntaps, p.dtype, p.hi_off, p.wi_off = 1, 9, e.dt, -1, -1
    p.osN, p.osH, p.osW = x.H * x.W * Cc, x.W * Cc, Cc
    p.out = 4096
    return e.lib.dp_conv2d_kernel_class(C.byref(p))


def act(n, h, wd):
    x = torch.randn((n, h, wd, Cc), generator=g).to(tdt)
    return Act(x.cuda(), n, h, wd, Cc), x


if os.environ.get("CHECK", "1") != "0":
    bad = 0
    for (n, h, wd, relu) in [(1, 13, 21, True), (2, 9, 17, False), (3, 37, 45, True), (1, 8, 16, True), (5, 6, 16, False), (2, 25, 42, True), (1, 50, 84, True),
                             (2, 64, 35, False), (1, 47, 130, True), (8, 13, 32, True), (1, 4, 40, True), (2, 3, 50, True), (1, 1, 130, False)]:
        xa, x = act(n, h, wd)
        L.set_policy("conv_wsq", 1)
        k = klass(xa)
        got = e.conv(layer, xa, relu=relu).t.float().cpu()
        ref = F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
        ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1)
        d = (got.double() - ref).abs()
        ok = bool((d <= ulp * ref.abs() + 2e-3).all())
        # image alone == image in the batch, bit for bit
        same = True
        for i in range(n):
            one = e.conv(layer, Act(xa.t[i:i + 1].contiguous(), 1, h, wd, Cc), relu=relu).t.float().cpu()
            same = same and torch.equal(one[0], got[i])
        print("check N=%d %dx%d relu=%d: class %d  max |err| %.3e (allowed %.3e at the largest value)  %s  batch==single %s" % (
            n, h, wd, relu, k, float(d.max()), float(ulp * ref.abs().max() + 2e-3), "ok" if ok else "WRONG", same))
        bad += (not ok) + (not same)
    # large maps: against the older kernels (verified against torch by their own tests), two rounding steps
    for (n, h, wd) in [(2, 200, 336), (3, 100, 168)]:
        xa, x = act(n, h, wd)
        L.set_policy("conv_wsq", 1)
        got = e.conv(layer, xa, relu=True).t.float()
        L.set_policy("conv_wsq", 0)
        want = e.conv(layer, xa, relu=True).t.float()
        L.set_policy("conv_wsq", 1)
        d = (got - want).abs()
        ok = bool((d <= 2 * ulp * want.abs() + 1e-3).all())
        nz = float((d > 0).float().mean())
        print("check N=%d %dx%d against class %d: max |diff| %.3e, %.4f of the elements differ  %s" % (n, h, wd, 6, float(d.max()), nz, "ok" if ok else "WRONG"))
        bad += not ok
    print("CHECK", "PASSED" if bad == 0 else "FAILED (%d)" % bad)


def bench(xa, out, reps=40, passes=5):
    for _ in range(5):
        e.conv(layer, xa, relu=True, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            e.conv(layer, xa, relu=True, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "200x336,100x168,50x84,25x42,13x21").split(",")]
# warm the clocks
xa, _ = act(N, 100, 168)
out = torch.empty((N, 100, 168, Cc), dtype=tdt, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    e.conv(layer, xa, relu=True, out=out)
torch.cuda.synchronize()
for (h, wd) in shapes:
    xa, _ = act(N, h, wd)
    out = torch.empty((N, h, wd, Cc), dtype=tdt, device="cuda")
    fl = 2.0 * N * h * wd * Cc * Cc * 9
    res = []
    for name, pol in (("wsq (class 10)", {"conv_wsq": 1}), ("wsr (class 6)", {"conv_wsq": 0}), ("ring", {"conv_wsq": 0, "conv_ws": 0})):
        with L.policy(**pol):
            k = klass(xa)
            ms = bench(xa, out)
        res.append("%s: class %d %.4f ms %.0f TFLOP/s (%.3f)" % (name, k, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500))
    print("N=%d %dx%d  " % (N, h, wd) + " | ".join(res))

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


INPUT = os.environ.get("INPUT", "randn")   # randn | relu (half of the values zero, like a layer behind a ReLU) | zero


def act(n, h, wd):
    x = torch.randn((n, h, wd, Cc), generator=g)
    x = (torch.relu(x) if INPUT == "relu" else x * 0 if INPUT == "zero" else x).to(tdt)
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
    # the host's scheduling hints change the split of the steps over workgroups, never the bits
    for shp in (16, 32):
        L.set_policy("wsq_shape", shp)
        for (n, h, wd) in [(6, 64, 100), (6, 32, 50), (6, 16, 25), (1, 16, 25), (2, 50, 84), (8, 25, 42), (3, 100, 168), (1, 64, 100)]:
            xa, x = act(n, h, wd)
            outs = []
            for hint in (0, 1, 2):
                e._shared_chip = hint
                outs.append(e.conv(layer, xa, relu=True).t.clone())
            e._shared_chip = 0
            one = torch.cat([e.conv(layer, Act(xa.t[i:i + 1].contiguous(), 1, h, wd, Cc), relu=True).t for i in range(n)])
            torch.cuda.synchronize()
            res = [torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]), torch.equal(outs[0], one)]
            if not all(res):
                d = (outs[0].float() - outs[1].float()).abs() + (outs[0].float() - outs[2].float()).abs() + (outs[0].float() - one.float()).abs()
                idx = torch.nonzero(d.sum(-1) > 0)
                print("  differing pixels (n, y, x):", idx[:12].tolist(), "count", idx.shape[0], "channels of the first:", torch.nonzero(d[tuple(idx[0])] > 0).flatten()[:8].tolist())
            print("hints shape=%d N=%d %dx%d: hint1 == hint0 %s, hint2 == hint0 %s, single == batch %s" % (shp, n, h, wd, *res))
            bad += not all(res)
    L.set_policy("wsq_shape", 16)
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
    for name, pol in (("ws1 16x16x32", {"conv_wsq": 1, "wsq_shape": 16}), ("wsq 32x32x16", {"conv_wsq": 1, "wsq_shape": 32}), ("wsr", {"conv_wsq": 0}), ("ring", {"conv_wsq": 0, "conv_ws": 0})):
        with L.policy(**pol):
            k = klass(xa)
            ms = bench(xa, out)
        res.append("%s: %d %.4f ms %.0f TF (%.3f)" % (name, k, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500))
    print("N=%d %dx%d input=%s  " % (N, h, wd, INPUT) + " | ".join(res))

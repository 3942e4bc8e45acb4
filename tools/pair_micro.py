"""conv3 + residual + ReLU -> next conv1 + ReLU as one launch (dp_bottleneck_pair_nhwc) against the two separate launches, res3 / res4 shapes.
usage: python tools/pair_micro.py [batch] [dtype]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Act, Engine
from densepose_torchscript_amd.pack import conv_from_oihw

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dt]
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype=dt)
g = torch.Generator().manual_seed(0)


def bench(fn, reps=50, passes=5):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(passes):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


for Cm, H, W in ((256, 50, 84), (128, 100, 168)):
    Co = 4 * Cm
    mk = lambda co, ci: (torch.randn((co, ci, 1, 1), generator=g) * (1.0 / ci) ** 0.5).numpy()  # noqa: E731
    l3 = conv_from_oihw("conv3", mk(Co, Cm), np.zeros(Co, np.float32), Cm, 1, 0, 1, e.dt, e.device)
    l1 = conv_from_oihw("conv1n", mk(Cm, Co), np.zeros(Cm, np.float32), Co, 1, 0, 1, e.dt, e.device)
    ta = Act(torch.relu(torch.randn((N, H, W, Cm), generator=g)).to(tdt).cuda(), N, H, W, Cm)
    ra = Act(torch.randn((N, H, W, Co), generator=g).to(tdt).cuda(), N, H, W, Co)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        e.bottleneck_pair(l3, l1, ta, ra)
    torch.cuda.synchronize()
    tp = bench(lambda: e.bottleneck_pair(l3, l1, ta, ra))
    t3 = bench(lambda: e.conv(l3, ta, relu=True, residual=ra))
    x = e.conv(l3, ta, relu=True, residual=ra)
    t1 = bench(lambda: e.conv(l1, x, relu=True))
    mb = N * H * W * 2 * (Cm + 2 * Co + Cm) / 1e6
    fl = 2.0 * N * H * W * 2 * Cm * Co
    print("N=%d %dx%d %d->%d->%d: pair %.1f us (%.2f TB/s of %.0f MB, %.0f TFLOP/s) | conv3 %.1f + conv1 %.1f = %.1f us" % (
        N, H, W, Cm, Co, Cm, tp * 1e3, mb / tp / 1e9 * 1e3, mb, fl / tp / 1e9, t3 * 1e3, t1 * 1e3, (t3 + t1) * 1e3))

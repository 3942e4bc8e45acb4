"""The fused stem (dp_stem_pool_nhwc: 7x7/2 conv + FrozenBN + ReLU + 3x3/2 max-pool, one launch) alone in a loop: A/B timing.
usage: python tools/stem_micro.py [N Hp Wp reps]     DP_HIP_LIB=<.so> loads another build (with DP_SKIP_STAMP_CHECK=1)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np
import torch

from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd import lib as L
from densepose_torchscript_amd.engine import Act, Engine
from densepose_torchscript_amd.pack import stem_paired_conv

N, Hp, Wp, reps = [int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else "8 800 1344 100".split())]
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
g = torch.Generator().manual_seed(0)
img = torch.randint(0, 256, (N, 3, Hp, Wp - 11), generator=g, dtype=torch.uint8).to(e.device)
Wq = Wp // 2 + 3
buf = torch.empty((N, Hp, Wq, 8), dtype=e.tdt, device=e.device)
p = L.PreprocessParams()
p.src, p.dst, p.paired = img.data_ptr(), buf.data_ptr(), 1
p.n_img, p.h, p.w, p.Hp, p.Wp, p.dtype = N, Hp, Wp - 11, Hp, Wp, e.dt
for i, (m, s) in enumerate(zip((103.53, 116.28, 123.675), (1.0, 1.0, 1.0))):
    p.mean[i], p.std[i] = m, s
L.check(e.lib.dp_preprocess_u8(C.byref(p), e._stream()), "dp_preprocess_u8")
wt = (torch.randn((64, 3, 7, 7), generator=g) * 0.01).to(torch.bfloat16).float()
b = torch.randn((64,), generator=g) * 20
layer = stem_paired_conv("stem", wt.numpy(), b.numpy(), e.dt, e.device)
xa = Act(buf, N, Hp, Wq, 8)
first = e.stem_pool(layer, xa).t.clone()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    e.stem_pool(layer, xa)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = e.stem_pool(layer, xa)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / reps)
ms = float(np.median(ts))
fl = 2.0 * N * (Hp // 2) * (Wp // 2) * 64 * 147
nb = buf.numel() * 2 + out.t.numel() * 2
print("stem %dx%dx%d lib=%s: %.4f ms  %.1f TF/s  %.2f TB/s  checksum %.6e  repeatable %s" % (
    N, Hp, Wp, os.path.basename(os.environ.get("DP_HIP_LIB", "default")), ms, fl / ms / 1e9, nb / ms / 1e9, float(first.float().sum()), torch.equal(first, out.t)))

"""The fused res2 bottleneck tail (dp_bottleneck_tail_nhwc) alone in a loop: A/B timing and rocprofv3 --pmc passes.
usage: tail_micro.py [N H W reps next]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
N, H, W, reps, nxt = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else "8 200 336 20 1".split())]
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
g = torch.Generator().manual_seed(0)
mk = lambda co, ci, k: (torch.randn((co, ci, k, k), generator=g) * (2.0 / (ci * k * k)) ** 0.5).numpy()
l2 = conv_from_oihw("c2", mk(64, 64, 3), np.zeros(64, np.float32), 64, 1, 1, 1, e.dt, e.device, plane_major=False)
l3 = conv_from_oihw("c3", mk(256, 64, 1), np.zeros(256, np.float32), 64, 1, 0, 1, e.dt, e.device)
l1 = conv_from_oihw("c1", mk(64, 256, 1), np.zeros(64, np.float32), 256, 1, 0, 1, e.dt, e.device)
t1 = Act(torch.randn((N, H, W, 64), generator=g).relu().to(torch.bfloat16).cuda(), N, H, W, 64)
res = Act(torch.randn((N, H, W, 256), generator=g).to(torch.bfloat16).cuda(), N, H, W, 256)
for _ in range(3): e.bottleneck_tail(l2, l3, l1 if nxt else None, t1, res)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): e.bottleneck_tail(l2, l3, l1 if nxt else None, t1, res)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
px = N * H * W
fl = 2.0 * px * (64 * 576 + 64 * 256 + (256 * 64 if nxt else 0))
nb = 2.0 * px * (64 + 256 + 256 + (64 if nxt else 0))
print("tail %dx%dx%d next=%d lib=%s: %.3f ms  %.1f TF/s  %.2f TB/s" % (N, H, W, nxt, os.path.basename(os.environ.get("DP_HIP_LIB", "default")), dt * 1e3, fl / dt / 1e12, nb / dt / 1e12))

"""One conv layer in a loop (for rocprofv3 --pmc passes and A/B timing)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
N, Cin, H, W, Cout, k = [int(x) for x in (sys.argv[1:7] if len(sys.argv) > 6 else "8 256 200 336 256 3".split())]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
use_res = len(sys.argv) > 8 and sys.argv[8] == "res"
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
g = torch.Generator().manual_seed(0)
w = (torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5).numpy()
layer = conv_from_oihw("micro", w, np.zeros(Cout, np.float32), Cin, 1, k // 2, 1, e.dt, e.device)
x = Act(torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda(), N, H, W, Cin)
out = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device="cuda")
res = Act(torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16).cuda(), N, H, W, Cout) if use_res else None
for _ in range(3): e.conv(layer, x, relu=True, out=out, residual=res)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): e.conv(layer, x, relu=True, out=out, residual=res)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * N * H * W * Cout * Cin * k * k
nb = 2.0 * N * H * W * (Cin + Cout * (2 if use_res else 1))
print("conv %dx%dx%dx%d -> %d k%d %s: %.3f ms  %.1f TF/s  %.2f TB/s" % (N, H, W, Cin, Cout, k, "+res" if use_res else "", dt * 1e3, fl / dt / 1e12, nb / dt / 1e12))

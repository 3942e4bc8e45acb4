"""One DensePose-head layer (3x3, 512 -> 512 on R x 28 x 28 ROI maps) in a loop: the row-streaming kernel (class 7) against the LDS-ring
kernel (DP_CONV_ROWS=0), HIP-event time per launch.  usage: rows_micro.py [R] [Cin] [H] [W] [dtype]"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ci = int(sys.argv[2]) if len(sys.argv) > 2 else 512
H = int(sys.argv[3]) if len(sys.argv) > 3 else 28
W = int(sys.argv[4]) if len(sys.argv) > 4 else 28
dt = sys.argv[5] if len(sys.argv) > 5 else "bf16"
Co = int(os.environ.get("COUT", "512"))
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype=dt)
g = torch.Generator().manual_seed(1)
x = torch.randn((R, H, W, Ci), generator=g).to(e.tdt).to(e.device)
w = (torch.randn((Co, Ci, 3, 3), generator=g) * (1.0 / (9 * Ci)) ** 0.5)
layer = conv_from_oihw("fcn", w.numpy(), torch.zeros(Co).numpy(), Ci, 1, 1, 1, e.dt, e.device)
xa = Act(x, R, H, W, Ci)
flops = 2.0 * R * H * W * Co * Ci * 9
def run(n=30):
    for _ in range(5): e.conv(layer, xa, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): e.conv(layer, xa, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mode in (os.environ.get("ROWS_ON", "1"), "0") * 2:
    os.environ["DP_CONV_ROWS"] = mode
    ms = run()
    print("R=%d %dx%d %d->%d %s DP_CONV_ROWS=%s: %.1f us  %.0f TFLOP/s (%.3f of 2.5 PF)" % (R, H, W, Ci, Co, dt, mode, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 2500))

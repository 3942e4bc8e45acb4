"""One DensePose-head layer (3x3, 512 -> 512 on R x 28 x 28 ROI maps) in a loop: the row-streaming kernel (class 7) against the LDS-ring
kernel, HIP-event time per launch.  usage: [MODES=rows32,rows16,ring,chain,rows32-lockstep] [COUT=512] rows_micro.py [R] [Cin] [H] [W] [dtype]"""
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
def run(n=100):
    for _ in range(10): e.conv(layer, xa, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): e.conv(layer, xa, relu=True)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
MODES = {"ring": {"DP_CONV_ROWS": "0"}, "rows16": {"DP_CONV_ROWS": "2", "DP_CONV_ROWS2": "0"}, "rows32": {"DP_CONV_ROWS": "2", "DP_CONV_ROWS2": "1", "DP_CONV_ROWS2_256": "1"},
         "rows32-lockstep": {"DP_CONV_ROWS": "2", "DP_CONV_ROWS2": "1", "DP_CONV_ROWS2_256": "1", "DP_CONV_ROWS2_LOCKSTEP": "1"}, "chain": {"DP_CONV_ROWS": "2", "DP_CONV_ROWS2": "0", "DP_CONV_ROWS_CHAIN": "1"}}
names = os.environ.get("MODES", "rows32,rows32-lockstep,rows16,ring").split(",")
def setmode(name):
    for k in ("DP_CONV_ROWS", "DP_CONV_ROWS2", "DP_CONV_ROWS2_LOCKSTEP", "DP_CONV_ROWS_CHAIN", "DP_CONV_ROWS2_256"):
        os.environ.pop(k, None)
    os.environ.update(MODES[name])
    from densepose_torchscript_amd import lib as _L; _L.apply_env_policy()   # the library reads no environment: the host applies it
# the chip's clock settles over hundreds of milliseconds: 1.5 s of the same load first, then the modes in turn, three passes
setmode(names[0])
import time
t0 = time.time()
while time.time() - t0 < 1.5:
    run(50)
res = {n: [] for n in names}
for _ in range(3):
    for name in names:
        setmode(name)
        res[name].append(run())
for name in names:
    ms = sorted(res[name])[1]
    print("R=%d %dx%d %d->%d %s %-16s median %.1f us (%s)  %.0f TFLOP/s (%.3f of 2.5 PF)" % (R, H, W, Ci, Co, dt, name, ms * 1e3, " ".join("%.1f" % (x * 1e3) for x in res[name]), flops / ms / 1e9, flops / ms / 1e9 / 2500))

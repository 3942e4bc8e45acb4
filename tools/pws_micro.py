"""The long-K pointwise layers of the trunk / FPN / box head, one layer in a loop: the weight-stationary pointwise kernel (dp_conv_pw.hip,
kernel class 9) against what the layer ran on before (policy key conv_pws = 0: LDS-ring / streaming kernels). HIP-event time per launch,
1.5 s of warm-up, the modes in turn, median of three passes of 100 launches.   usage: pws_micro.py [batch] [dtype]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state, lib as L
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype=dt)
LAYERS = [  # name, N, H, W, Cin, Cout, relu, residual mode
    ("res4 conv1", B, 50, 84, 1024, 256, True, None),
    ("fpn_lateral4", B, 50, 84, 1024, 256, False, "up"),
    ("fc2", 1, 1, 1000 * B, 1024, 1024, True, None),
    ("res5 conv1", B, 25, 42, 2048, 512, True, None),
    ("res5 conv3", B, 25, 42, 512, 2048, True, "lin"),
    ("fpn_lateral5", B, 25, 42, 2048, 256, False, None),
    ("fpn_lateral3", B, 100, 168, 512, 256, False, "up"),
    ("res4 conv3", B, 50, 84, 256, 1024, True, "lin"),
]
only = os.environ.get("LAYERS")
g = torch.Generator().manual_seed(1)
for name, N, H, W, Ci, Co, relu, rmode in LAYERS:
    if only and name not in only.split(","):
        continue
    x = torch.randn((N, H, W, Ci), generator=g).to(e.tdt).to(e.device)
    w = torch.randn((Co, Ci, 1, 1), generator=g) * (1.0 / Ci) ** 0.5
    layer = conv_from_oihw("pw", w.numpy(), torch.zeros(Co).numpy(), Ci, 1, 0, 1, e.dt, e.device)
    xa = Act(x, N, H, W, Ci)
    ra, rshift = None, 0
    if rmode == "lin":
        ra = Act(torch.randn((N, H, W, Co), generator=g).to(e.tdt).to(e.device), N, H, W, Co)
    elif rmode == "up":
        ra, rshift = Act(torch.randn((N, H // 2, W // 2, Co), generator=g).to(e.tdt).to(e.device), N, H // 2, W // 2, Co), 1
    out = torch.empty((N, H, W, Co), dtype=e.tdt, device=e.device)
    M = N * H * W
    flops = 2.0 * M * Co * Ci
    nbytes = 2.0 * (M * Ci + M * Co + Co * Ci + (ra.t.numel() if ra is not None else 0))

    def run(n=100):
        for _ in range(10):
            e.conv(layer, xa, relu=relu, residual=ra, rshift=rshift, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            e.conv(layer, xa, relu=relu, residual=ra, rshift=rshift, out=out)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    modes = {"pws": (1, 0), "before": (0, 0)}
    if os.environ.get("SKEW"):       # `make exp` builds: the skewed schedule too
        modes["pws-skew"] = (1, 1)
    L.set_policy("conv_pws", 1)
    t0 = time.time()
    while time.time() - t0 < 1.5:
        run(50)
    res = {m: [] for m in modes}
    for _ in range(3):
        for m, v in modes.items():
            L.set_policy("conv_pws", v[0]); L.set_policy("pws_skew", v[1])
            res[m].append(run())
    for m in modes:
        ms = sorted(res[m])[1]
        print("%-13s %dx%dx%d %4d->%-4d %s %-12s median %6.1f us (%s)  %5.0f TFLOP/s (%.3f)  %.2f TB/s algorithmic" % (
            name, N, H, W, Ci, Co, dt, m, ms * 1e3, " ".join("%.1f" % (v * 1e3) for v in res[m]), flops / ms / 1e9, flops / ms / 1e9 / 2500, nbytes / ms / 1e9), flush=True)
L.reset_policy()

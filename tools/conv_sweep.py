"""Time every conv variant on the real layer shapes (batch 8) to calibrate choose_conv_kernel."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
shapes = [  # N, Cin, H, W, Cout, k, name
    (8, 256, 200, 336, 256, 3, "p2 3x3"), (8, 256, 100, 168, 256, 3, "p3 3x3"), (8, 256, 50, 84, 256, 3, "p4 3x3"),
    (8, 256, 25, 42, 256, 3, "p5 3x3"), (8, 128, 100, 168, 128, 3, "res3 conv2"), (8, 512, 25, 42, 512, 3, "res5 conv2"),
    (64, 512, 28, 28, 512, 3, "dp head"), (32, 512, 28, 28, 512, 3, "dp head R=32"), (64, 256, 28, 28, 512, 3, "dp fcn1"),
    (8000, 12544, 1, 1, 1024, 1, "fc1"), (8000, 1024, 1, 1, 1024, 1, "fc2"),
    (8, 256, 200, 336, 256, 1, "lateral2/dec pred"), (8, 512, 100, 168, 256, 1, "lateral3"), (8, 1024, 50, 84, 256, 1, "lateral4"),
    (8, 2048, 25, 42, 256, 1, "lateral5"), (8, 512, 100, 168, 128, 1, "res3 conv1"), (8, 128, 100, 168, 512, 1, "res3 conv3"),
    (8, 1024, 50, 84, 256, 1, "res4 conv1"), (8, 256, 50, 84, 1024, 1, "res4 conv3"), (8, 2048, 25, 42, 512, 1, "res5 conv1"),
    (8, 512, 25, 42, 2048, 1, "res5 conv3"), (4, 256, 200, 336, 256, 3, "p2 3x3 half batch"), (4, 256, 100, 168, 256, 3, "p3 half"),
]
g = torch.Generator().manual_seed(0)
for N, Cin, H, W, Cout, k, name in shapes:
    w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).numpy()
    layer = conv_from_oihw("m", w, np.zeros(Cout, np.float32), Cin, 1, k // 2, 1, e.dt, e.device)
    x = Act(torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda(), N, H, W, Cin)
    out = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    res = []
    for force in ("0", "1", "3", "2"):
        os.environ["DP_CONV_BIG"] = force
        from densepose_torchscript_amd import lib as _L; _L.apply_env_policy()
        for _ in range(2): e.conv(layer, x, relu=True, out=out)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): e.conv(layer, x, relu=True, out=out)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
    fl = 2.0 * N * H * W * Cout * Cin * k * k
    M = N * H * W
    print("%-20s M=%-7d K=%-6d N=%-5d generic %.3f  ring256 %.3f  ring256x128 %.3f  ring128 %.3f ms   best %s %.0f TF/s" % (
        name, M, Cin * k * k, Cout, res[0], res[1], res[2], res[3], ("generic", "ring256", "ring256x128", "ring128")[int(np.argmin(res))], fl / min(res) / 1e9))

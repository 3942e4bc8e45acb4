"""Tile height of the 256-cout LDS-ring kernel (DP_CONV_TP = 4 .. 8, i.e. 128 .. 256 pixel rows) against choose_ring256_tp's pick on the layers
that run on it, warm clocks, median of 3 x 50 launches.  usage: tp_sweep.py [batch]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
shapes = [(B, 256, 200, 336, 256, 3, "rpn p2 (no head)"), (B, 256, 100, 168, 256, 3, "rpn p3"), (B, 256, 50, 84, 256, 3, "rpn p4"),
          (B, 512, 100, 168, 256, 1, "lateral3"), (B, 1024, 50, 84, 256, 1, "lateral4 / res4 conv1"), (B, 2048, 25, 42, 512, 1, "res5 conv1"),
          (B, 512, 25, 42, 2048, 1, "res5 conv3"), (1000 * B, 1024, 1, 1, 1024, 1, "fc2"), (1000 * B, 12544, 1, 1, 1024, 1, "fc1 (unsplit)")]
g = torch.Generator().manual_seed(0)
def run(layer, x, out, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): e.conv(layer, x, relu=True, out=out)
    e0.record()
    for _ in range(n): e.conv(layer, x, relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
os.environ["DP_CONV_WS"] = "0"; os.environ["DP_CONV_BIG"] = "1"
from densepose_torchscript_amd import lib as _L; _L.apply_env_policy()   # the library reads no environment
for N, Cin, H, W, Cout, k, name in shapes:
    w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).numpy()
    layer = conv_from_oihw("m", w, np.zeros(Cout, np.float32), Cin, 1, k // 2, 1, e.dt, e.device)
    x = Act(torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda(), N, H, W, Cin)
    out = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 0.7: run(layer, x, out, 20)
    var = ["auto", "4", "5", "6", "7", "8"]
    res = {v: [] for v in var}
    for _ in range(3):
        for v in var:
            os.environ.pop("DP_CONV_TP", None)
            if v != "auto": os.environ["DP_CONV_TP"] = v
            _L.apply_env_policy()
            res[v].append(run(layer, x, out, 50))
    os.environ.pop("DP_CONV_TP", None)
    med = {v: sorted(t)[1] for v, t in res.items()}
    best = min(med, key=med.get)
    flag = "" if med["auto"] <= 1.03 * med[best] else "   <-- auto %.0f %% slower than TP=%s" % (100 * (med["auto"] / med[best] - 1), best)
    print("%-24s M=%-7d K=%-6d N=%-5d  " % (name, N * H * W, Cin * k * k, Cout) + "  ".join("%s %.1f" % (v, med[v]) for v in var) + " us" + flag)

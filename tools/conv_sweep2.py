"""The default kernel policy against every forced kernel class on the real layer shapes, warmed up (1 s of load first, then 3 passes of 50
launches per variant, median): which layers choose_conv_kernel sends to a slower class.  usage: conv_sweep2.py [batch]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
from densepose_torchscript_amd import lib as _L   # the library reads no environment: apply_env_policy() after every change
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
R = 8 * B
shapes = [  # N, Cin, H, W, Cout, k, name
    (B, 256, 200, 336, 256, 3, "p2 3x3"), (B, 256, 100, 168, 256, 3, "p3 3x3"), (B, 256, 50, 84, 256, 3, "p4 3x3"),
    (B, 256, 25, 42, 256, 3, "p5 3x3"), (B, 256, 13, 21, 256, 3, "p6 3x3"), (B, 128, 100, 168, 128, 3, "res3 conv2"), (B, 512, 25, 42, 512, 3, "res5 conv2"),
    (R, 512, 28, 28, 512, 3, "dp head"), (R, 256, 28, 28, 512, 3, "dp fcn1"),
    (1000 * B, 12544, 1, 1, 1024, 1, "fc1"), (1000 * B, 1024, 1, 1, 1024, 1, "fc2"),
    (B, 256, 200, 336, 256, 1, "lateral2/dec pred"), (B, 512, 100, 168, 256, 1, "lateral3"), (B, 1024, 50, 84, 256, 1, "lateral4"),
    (B, 2048, 25, 42, 256, 1, "lateral5"), (B, 512, 100, 168, 128, 1, "res3 conv1"), (B, 128, 100, 168, 512, 1, "res3 conv3"),
    (B, 1024, 50, 84, 256, 1, "res4 conv1"), (B, 256, 50, 84, 1024, 1, "res4 conv3"), (B, 2048, 25, 42, 512, 1, "res5 conv1"),
    (B, 512, 25, 42, 2048, 1, "res5 conv3"), (B, 64, 200, 336, 64, 1, "res2.0 conv1"), (B, 64, 200, 336, 256, 1, "res2 conv3-type"),
]
VAR = {"default": {}, "generic": {"DP_CONV_BIG": "0"}, "ring256": {"DP_CONV_BIG": "1"}, "ring256x128": {"DP_CONV_BIG": "3"}, "ring128": {"DP_CONV_BIG": "2"},
       "stream": {"DP_CONV_BIG": "5"}, "no-ws/rows/pws": {"DP_CONV_WS": "0", "DP_CONV_ROWS": "0", "DP_CONV_PWS": "0"}}
KEYS = ("DP_CONV_BIG", "DP_CONV_WS", "DP_CONV_ROWS", "DP_CONV_PWS")
g = torch.Generator().manual_seed(0)
def run(layer, x, out, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5): e.conv(layer, x, relu=True, out=out)
    e0.record()
    for _ in range(n): e.conv(layer, x, relu=True, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, Cin, H, W, Cout, k, name in shapes:
    w = (torch.randn((Cout, Cin, k, k), generator=g) * 0.05).numpy()
    layer = conv_from_oihw("m", w, np.zeros(Cout, np.float32), Cin, 1, k // 2, 1, e.dt, e.device)
    x = Act(torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).cuda(), N, H, W, Cin)
    out = torch.empty((N, H, W, Cout), dtype=torch.bfloat16, device="cuda")
    t0 = time.time()
    while time.time() - t0 < 0.7: run(layer, x, out, 20)
    res = {v: [] for v in VAR}
    for _ in range(3):
        for v, env in VAR.items():
            for kk in KEYS: os.environ.pop(kk, None)
            os.environ.update(env)
            _L.apply_env_policy()
            res[v].append(run(layer, x, out, 50))
    for kk in KEYS: os.environ.pop(kk, None)
    _L.apply_env_policy()
    med = {v: sorted(t)[1] for v, t in res.items()}
    best = min(med, key=med.get)
    fl = 2.0 * N * H * W * Cout * Cin * k * k
    flag = "" if med["default"] <= 1.03 * med[best] else "   <-- default %.0f %% slower than %s" % (100 * (med["default"] / med[best] - 1), best)
    print("%-18s M=%-7d K=%-6d N=%-5d  " % (name, N * H * W, Cin * k * k, Cout) + "  ".join("%s %.1f" % (v, med[v]) for v in VAR) + "  us   (default %.0f TF/s)%s" % (fl / med["default"] / 1e6, flag))

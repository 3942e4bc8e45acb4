"""The weight-stationary 3x3 kernel against the LDS-ring kernel on small maps (bit for bit): usage wsr_small_check.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densepose_torchscript_amd import TINY_OPTS, get_config, make_synthetic_state
from densepose_torchscript_amd.engine import Engine, Act
from densepose_torchscript_amd.pack import conv_from_oihw
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", TINY_OPTS)
e = Engine(cfg, make_synthetic_state(cfg, 0), dtype="bf16")
g = torch.Generator().manual_seed(1)
for C in (256, 128):
    w = torch.randn((C, C, 3, 3), generator=g) * (1.0 / (9 * C)) ** 0.5
    layer = conv_from_oihw("l", w.numpy(), (torch.randn(C, generator=g) * 0.1).numpy(), C, 1, 1, 1, e.dt, e.device)
    for (N, H, W) in [(3, 8, 13), (1, 13, 21), (1, 25, 42), (3, 16, 25), (2, 7, 9), (1, 6, 16), (2, 9, 17), (1, 12, 33), (5, 8, 13), (1, 50, 84)]:
        x = Act(torch.randn((N, H, W, C), generator=g).to(e.tdt).to(e.device), N, H, W, C)
        outs = {}
        for ws in ("1", "0"):
            os.environ["DP_CONV_WS"] = ws
            os.environ["DP_WS_MIN_M"] = "1"
            from densepose_torchscript_amd import lib as _L; _L.apply_env_policy()
            outs[ws] = e.conv(layer, x, relu=True).t.clone()
        torch.cuda.synchronize()
        d = (outs["1"].float() - outs["0"].float()).abs()
        print(C, (N, H, W), "equal" if torch.equal(outs["1"], outs["0"]) else "DIFFER max %.4g at %s" % (float(d.max()), str(torch.nonzero(d == d.max())[0].tolist())))

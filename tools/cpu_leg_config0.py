"""BASELINE.json configs[0]: densepose_rcnn_R_50_FPN_s1x_legacy, 1 x 800 x 1333 fp32 on the CPU (the reference's plumbing case) -
the oracle (oracle/ref_cpu.py, shown equal to the imported reference in tests/test_oracle_vs_reference.py) timed on this box's host
cores. usage: cpu_leg_config0.py [threads] [frames]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from densepose_torchscript_amd import get_config, make_synthetic_state
from oracle.ref_cpu import OracleModel
threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(os.cpu_count() or 1, 32)
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.set_num_threads(threads)
cfg = get_config("densepose_rcnn_R_50_FPN_s1x_legacy", ["TEST.DETECTIONS_PER_IMAGE", 8])
model = OracleModel(cfg, make_synthetic_state(cfg, 0))
imgs = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)) for i in range(4)]
for i in range(2): model(imgs[i])
ts = []
for i in range(frames):
    t0 = time.time(); out = model(imgs[i % 4]); ts.append(time.time() - t0)
print("densepose_rcnn_R_50_FPN_s1x_legacy 1x800x1333 fp32, oracle/ref_cpu.py on %d host threads (torch %s): %.3f images/s, p50 %.0f ms/img, R = %d"
      % (threads, torch.__version__, len(ts) / sum(ts), 1e3 * float(np.median(ts)), out["scores"].shape[0]))

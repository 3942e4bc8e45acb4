"""Host-side cost of ONE synchronous single-frame call (what run.py's loop sees): wall per call, host return time, cProfile of 200 calls."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cProfile
import pstats

import numpy as np
import torch

from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.predictor import DensePosePredictor

cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 0), dtype="bf16", resize="device", num_streams=2, use_graphs=True)
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(4)]
for i in range(8):
    pred(frames[i % 4])
torch.cuda.synchronize()
hs, ws = [], []
for i in range(50):
    t0 = time.perf_counter()
    pred(frames[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(t1 - t0)
    ws.append(t2 - t0)
print("single frame: host returns after %.3f ms, GPU done after %.3f ms (medians of 50)" % (1e3 * np.median(hs), 1e3 * np.median(ws)))
pr = cProfile.Profile()
pr.enable()
for i in range(200):
    pred(frames[i % 4])
    torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

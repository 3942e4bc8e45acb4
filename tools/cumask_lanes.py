"""Experiment: the two pipeline lanes of the default schedule on DISJOINT halves of the chip (HIP streams created with a CU mask) against
the same lanes sharing all CUs. MFMA-bound kernels run against the power limit and HBM-bound ones leave the matrix pipes idle: two
half-chip lanes might overlap the two kinds where time-slicing a full chip cannot (persistent one-workgroup-per-CU kernels never co-reside).
usage: python tools/cumask_lanes.py [eager|graphs]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.predictor import DensePosePredictor

mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
state = make_synthetic_state(cfg, 0)
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(8)]
pred = DensePosePredictor(cfg, state, dtype="bf16", resize="device", use_graphs=(mode == "graphs"), pipeline_depth=2)
if mode == "eager":
    pred.engine.overlap_decoder = False
    pred.engine.fork_levels = 0


def rate(seconds=2.0):
    for _ in range(6):
        pred.predict_batch(frames)
    pred.join()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        pred.predict_batch(frames)
        n += 1
    pred.join()
    torch.cuda.synchronize()
    return 8 * n / (time.perf_counter() - t0)


print("%s, lanes share the chip:            %.1f images/s" % (mode, rate()))
full = [0xffffffff] * 8
for name, m0, m1 in (("halves (CUs 0-127 | 128-255 of the mask)", [0xffffffff] * 4 + [0] * 4, [0] * 4 + [0xffffffff] * 4),
                     ("alternate bits (0x55.. | 0xaa..)", [0x55555555] * 8, [0xaaaaaaaa] * 8),
                     ("3/4 | 3/4 overlapping (lane 0: words 0-5, lane 1: words 2-7)", [0xffffffff] * 6 + [0] * 2, [0] * 2 + [0xffffffff] * 6)):
    pred.join()
    torch.cuda.synchronize()
    pred._lanes = [masked_stream(m0), masked_stream(m1)]
    pred._next_lane = 0
    pred.engine._graphs.clear()
    print("%s, %s: %.1f images/s" % (mode, name, rate()))

"""The CPU baseline (oracle/ref_cpu.py, the port bench.py times) at several thread counts on this box's host cores, incl. nproc: bench.py
used to cap the baseline at 32 threads (round 4; bench.py now uses the count that serves it best: 16) because torch's CPU convolutions stop scaling - and then collapse - beyond that on this class of host;
this prints the evidence (profiles/r5_cpu_threads.txt).   usage: cpu_threads.py [frames per setting]"""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from densepose_torchscript_amd import get_config, make_synthetic_state
from oracle.ref_cpu import OracleModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
model = OracleModel(cfg, make_synthetic_state(cfg, 0))
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)) for i in range(4)]
nproc = os.cpu_count() or 1
print("host: %d logical CPUs, torch %s" % (nproc, torch.__version__), flush=True)
for th in sorted({8, 16, 32, 64, nproc}):
    if th > nproc:
        continue
    torch.set_num_threads(th)
    model(frames[0])
    ts = []
    for i in range(n):
        t0 = time.time(); model(frames[i % 4]); ts.append(time.time() - t0)
    print("threads %4d: %.3f images/s (p50 %.0f ms over %d frames, 800x1333, R_50_FPN_s1x, R = 8, fp32)" % (th, 1.0 / float(np.median(ts)), 1e3 * float(np.median(ts)), n), flush=True)

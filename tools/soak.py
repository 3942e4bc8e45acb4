"""Soak test of the default schedule (HIP-graph replay, side-stream decoder, two pipeline lanes): the same batch, many
times, every result compared bit for bit with the first one. A race between lanes / streams / graph memory shows up as
a mismatch."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.predictor import DensePosePredictor
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 0), dtype="bf16", resize="device", use_graphs=True, pipeline_depth=2)
sets = [[torch.from_numpy(np.random.default_rng(1234 + 8 * s + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(8)]
        for s in range(2)]
ref = []
for s in range(2):
    r = pred.predict_batch(sets[s]); pred.join(); torch.cuda.synchronize()
    ref.append([{k: v.clone() for k, v in o.items()} for o in r])
bad = 0
t0 = time.perf_counter()
window = []
for it in range(steps):
    s = it & 1                       # alternate two different batches so that a stale buffer cannot go unnoticed
    window.append((s, pred.predict_batch(sets[s])))
    if len(window) > 2:              # compare a result two calls later (both lanes have moved on meanwhile)
        s0, out = window.pop(0)
        pred.join()
        for a, b in zip(ref[s0], out):
            for k in a:
                if not torch.equal(a[k], b[k].to(a[k].device)):
                    bad += 1
pred.join(); torch.cuda.synchronize()
for s0, out in window:               # the last two batches, still in the window when the loop ends
    for a, b in zip(ref[s0], out):
        for k in a:
            if not torch.equal(a[k], b[k].to(a[k].device)):
                bad += 1
dt = time.perf_counter() - t0
print("soak: %d batches, %d mismatching tensors, %.1f img/s (with the comparisons in the loop)" % (steps, bad, steps * 8 / dt))
sys.exit(1 if bad else 0)

"""Host-side timeline of one batch step (where the submitting thread spends its time): graph replays, count waits, phase B."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.predictor import DensePosePredictor
from densepose_torchscript_amd.engine import Engine
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 0), dtype="bf16", resize="device", num_streams=streams, use_graphs=True)
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(8)]
for _ in range(4): pred.predict_batch(frames)
torch.cuda.synchronize()
marks = []
def wrap(name):
    f = getattr(Engine, name)
    def g(self, *a, **k):
        t0 = time.perf_counter(); r = f(self, *a, **k); marks.append((name, t0, time.perf_counter())); return r
    setattr(Engine, name, g)
for n in ("_phase_a_run", "_phase_b", "densepose_branch"): wrap(n)
import densepose_torchscript_amd.predictor as P
rg = P.DensePosePredictor._resize_group
def rg2(self, chws):
    t0 = time.perf_counter(); r = rg(self, chws); marks.append(("resize", t0, time.perf_counter())); return r
P.DensePosePredictor._resize_group = rg2
ev_sync = torch.cuda.Event.synchronize
def es(self):
    t0 = time.perf_counter(); r = ev_sync(self); marks.append(("  event.synchronize", t0, time.perf_counter())); return r
torch.cuda.Event.synchronize = es
for it in range(3):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); pred.predict_batch(frames); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("step: host return %.2f ms, gpu done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    for n, a, b in sorted(marks, key=lambda m: m[1]): print("   %-22s start %6.2f  dur %6.2f ms" % (n, (a - t0) * 1e3, (b - a) * 1e3))
# back-to-back steps (the bench loop)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): pred.predict_batch(frames)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("10 steps: host %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))

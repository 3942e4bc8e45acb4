"""Per-layer table of the conv launches in one batch step (events on the launch stream)."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.options import EngineOptions
from densepose_torchscript_amd.predictor import DensePosePredictor
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 0), dtype=dtype, resize="device", num_streams=1, options=EngineOptions.from_env())
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(batch)]
for _ in range(2): pred.predict_batch(frames)
torch.cuda.synchronize()
eng = pred.engine
eng.overlap_decoder = False
eng.prof = []
t0 = time.perf_counter(); pred.predict_batch(frames); torch.cuda.synchronize(); wall = time.perf_counter() - t0
rows = {}
for cls, flops, e0, e1, name, nb in eng.prof:
    r = rows.setdefault(name, [0, 0.0, 0, cls, 0]); r[0] += flops; r[1] += e0.elapsed_time(e1); r[2] += 1; r[4] += nb
eng.prof = None
tot = sum(r[1] for r in rows.values())
print("wall %.2f ms, conv total %.2f ms, %.1f GF, %.1f MB (=%.2f ms at 8 TB/s)" % (wall * 1e3, tot, sum(r[0] for r in rows.values()) / 1e9, sum(r[4] for r in rows.values()) / 1e6, sum(r[4] for r in rows.values()) / 8e9))
for name, (fl, ms, n, cls, nb) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%-62s %s n=%d %8.3f ms %7.1f GF %7.1f TF/s %7.1f MB %6.0f GB/s" % (name[-62:], cls[-12:], n, ms, fl / 1e9, fl / ms / 1e9, nb / 1e6, nb / ms / 1e6))
# non-conv time: whole-step stage timing with events
ev = lambda: torch.cuda.Event(enable_timing=True)
import densepose_torchscript_amd.engine as E

import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from densepose_torchscript_amd import get_config, make_synthetic_state
from densepose_torchscript_amd.predictor import DensePosePredictor
cfg = get_config("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8])
pred = DensePosePredictor(cfg, make_synthetic_state(cfg, 0), dtype="bf16", resize="device", num_streams=2, use_graphs=True)
frames = [torch.from_numpy(np.random.default_rng(1234 + i).integers(0, 256, (800, 1333, 3), dtype=np.uint8)).cuda() for i in range(8)]
for _ in range(4): pred.predict_batch(frames)
torch.cuda.synchronize()
import cProfile, pstats
hs, ws = [], []
for _ in range(10):
    t0 = time.perf_counter(); pred.predict_batch(frames); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append(t1 - t0); ws.append(t2 - t0)
print("host return %.2f ms, gpu done %.2f ms" % (1e3 * np.median(hs), 1e3 * np.median(ws)))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): pred.predict_batch(frames)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

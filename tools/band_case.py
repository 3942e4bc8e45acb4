"""One golden case in a 16-bit mode against the fp32 golden, detection by detection (box / score distance, largest IUV deviation per map
relative to the map's largest value) - what tests/test_gpu_e2e.py's band tests aggregate.   usage: band_case.py <case> [dtype]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from conftest import golden_case_inputs, load_golden
from densepose_torchscript_amd.options import EngineOptions
from densepose_torchscript_amd.predictor import DensePosePredictor
from test_gpu_e2e import IUV_KEYS, _label_agreement, _match_to_reference
name, dt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "bf16")
meta, z = load_golden(name)
cfg, state, img = golden_case_inputs(meta)
pred = DensePosePredictor(cfg, state, dtype=dt, options=EngineOptions.from_env())
out = {k: v.cpu() for k, v in pred(torch.from_numpy(img)).items()}
s = meta["iuv_stride"]
gb, gs, rb, rs = out["pred_boxes"].numpy(), out["scores"].numpy(), z["out/pred_boxes"], z["out/scores"]
print(name, dt, "R", len(gb), "/", len(rb), "fuse_sc_tail", pred.engine.fuse_sc_tail)
for i in range(len(rb)):
    d = np.abs(gb - rb[i]).max(axis=1)
    j = int(d.argmin())
    errs = {k.split("_")[-1]: (float(np.abs(out[k][j].numpy()[:, ::s, ::s] - z["out/" + k][i]).max()), float(np.abs(z["out/" + k][i]).max())) for k in IUV_KEYS}
    print("   ref %d -> %d: box %.3f px (%s) score %.4f (ref %.4f) " % (i, j, d[j], " ".join("%.1f" % v for v in rb[i]), abs(gs[j] - rs[i]), rs[i])
          + " ".join("%s %.3f/%.1f" % (k, a, b) for k, (a, b) in errs.items()))
print("    hits, iuv:", _match_to_reference(out, z, s, 1.5, 0.05), " label agreement: %.5f over %d px" % _label_agreement(out, z, 1.5))

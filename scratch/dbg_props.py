import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from conftest import load_golden, golden_case_inputs
from densepose_torchscript_amd.predictor import DensePosePredictor
meta, z = load_golden("tiny_r50_dl")
cfg, state, img = golden_case_inputs(meta)
pred = DensePosePredictor(cfg, state, dtype="fp32"); pred.engine.keep_intermediates = True
out = pred(torch.from_numpy(img)); torch.cuda.synchronize()
props, ps, pc = pred.engine.inter["proposals"]
n = int(pc[0]); a = props[0, :n].cpu().numpy(); b = z["stage/proposal_boxes"]; sa = ps[0, :n].cpu().numpy(); sb = z["stage/objectness_logits"]
bad = np.nonzero(np.abs(a - b).max(1) > 1e-3)[0]
print("n", n, "bad rows", bad)
for r in bad: print(r, a[r], sa[r], "| ref", b[r], sb[r])
# rpn head logits compare
heads = pred.engine.inter["rpn_heads"]
for i, h in enumerate(heads):
    got = h.t.cpu().numpy()[0]
    rl = z["stage/rpn_logits_%d" % i][0].transpose(1, 2, 0); rd = z["stage/rpn_deltas_%d" % i][0].transpose(1, 2, 0)
    print(i, "logit err", np.abs(got[..., :3] - rl).max(), "delta err", np.abs(got[..., 3:15] - rd).max())

import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import golden_case_inputs, load_golden
from densepose_torchscript_amd.predictor import DensePosePredictor
IUV_KEYS = ("pred_densepose_coarse_segm", "pred_densepose_fine_segm", "pred_densepose_u", "pred_densepose_v")
for name in ["tiny_r50_s1x_a", "full_r50_s1x_small", "full_r50_dl_p28", "tiny_r50_dl", "tiny_r101_dl_p28_video", "full_r50_s1x_800x1333"]:
    meta, z = load_golden(name)
    cfg, state, img = golden_case_inputs(meta)
    for dt in ("bf16", "fp16"):
        pred = DensePosePredictor(cfg, state, dtype=dt)
        out = {k: v.cpu() for k, v in pred(torch.from_numpy(img)).items()}
        s = meta["iuv_stride"]
        gb, gs, rb, rs = out["pred_boxes"].numpy(), out["scores"].numpy(), z["out/pred_boxes"], z["out/scores"]
        line = []
        for i in range(len(rb)):
            d = np.abs(gb - rb[i]).max(axis=1) if len(gb) else np.array([np.inf])
            j = int(d.argmin())
            errs = {k.split("_")[-1]: (float(np.abs(out[k][j].numpy()[:, ::s, ::s] - z["out/" + k][i]).max()), float(np.abs(z["out/" + k][i]).max())) for k in IUV_KEYS} if len(gb) else {}
            line.append("box %.3f score %.4f " % (d[j], abs(gs[j] - rs[i]) if len(gb) else -1) + " ".join("%s %.3f/%.1f" % (k, a, b) for k, (a, b) in errs.items()))
        print(name, dt, "R", len(gb), "/", len(rb)); [print("   ", l) for l in line]
        from test_gpu_e2e import _label_agreement
        print("    label agreement (matched detections, full resolution): %.5f over %d px" % _label_agreement(out, z, 1.5))
# the reference's own fp16 run (tests/golden/<case>__half.npz) against its fp32 run, and the engine's fp16 mode against both
from test_gpu_e2e import _match_to_reference
for name in ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl"]:
    meta, z = load_golden(name)
    _, zh = load_golden(name + "__half")
    cfg, state, img = golden_case_inputs(meta)
    ref_out = {k: torch.from_numpy(zh["out/" + k]) for k in IUV_KEYS + ("pred_boxes", "scores")}
    print(name, "reference-half vs reference-fp32 (hits, iuv):", _match_to_reference(ref_out, z, 1, 0.5, 0.02), "R", len(z["out/scores"]), len(zh["out/scores"]))
    pred = DensePosePredictor(cfg, state, dtype="fp16")
    out = {k: v.cpu() for k, v in pred(torch.from_numpy(img)).items()}
    print(name, "engine-fp16 vs reference-fp32:", _match_to_reference(out, z, meta["iuv_stride"], 0.5, 0.02))
    print(name, "engine-fp16 vs reference-half:", _match_to_reference(out, zh, meta["iuv_stride"], 0.5, 0.02))

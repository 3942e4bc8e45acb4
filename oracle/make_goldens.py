"""Generate tests/golden/*.npz by running the UNMODIFIED reference (/root/reference) on CPU.

Run in the build container only:   python -m oracle.make_goldens [--only NAME]

Every fixture holds: the case description (config name + override list + weight seed + image
seed/shape - all inputs are regenerated from seeds by the tests), the sha256 of the synthetic
weights, the reference's 8-key output dict, the visualiser's labels/uv per detection
(/root/reference/visualizer.py:10-56) and, for the "stages" cases, intermediate tensors captured
with forward hooks on the reference's modules. Nothing of the reference's source is stored.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from densepose_torchscript_amd.config import TINY_OPTS, get_config  # noqa: E402
from densepose_torchscript_amd.weights import make_synthetic_state, state_checksum  # noqa: E402
from oracle.ref_import import REFERENCE_ROOT, build_reference_predictor  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

TINY = TINY_OPTS + ["MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7, "TEST.DETECTIONS_PER_IMAGE", 3]

# name -> (config, opts, weight seed, image seed, (H, W), capture stages?, IUV subsample stride)
CASES = {
    "tiny_r50_s1x_a": ("densepose_rcnn_R_50_FPN_s1x", TINY, 0, 11, (96, 160), True, 1),
    "tiny_r50_s1x_b": ("densepose_rcnn_R_50_FPN_s1x", TINY, 0, 12, (250, 140), False, 1),
    "tiny_r50_legacy": ("densepose_rcnn_R_50_FPN_s1x_legacy", TINY, 1, 13, (120, 200), True, 1),
    "tiny_r101_s1x": ("densepose_rcnn_R_101_FPN_s1x", TINY, 2, 14, (128, 128), False, 1),
    "tiny_r50_dl": ("densepose_rcnn_R_50_FPN_DL_s1x", TINY, 3, 15, (96, 160), True, 1),
    "tiny_r101_dl": ("densepose_rcnn_R_101_FPN_DL_s1x", TINY, 4, 16, (300, 500), False, 1),
    # full-width network, reduced frame (CPU-oracle replay in a few seconds)
    "full_r50_s1x_small": ("densepose_rcnn_R_50_FPN_s1x",
                           ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 448, "TEST.DETECTIONS_PER_IMAGE", 4],
                           0, 21, (256, 400), False, 4),
    # BASELINE.json configs[2]: the deeper backbone at FULL channel width (33 bottleneck blocks: res4 x23), reduced frame
    "full_r101_s1x_small": ("densepose_rcnn_R_101_FPN_s1x",
                            ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 448, "TEST.DETECTIONS_PER_IMAGE", 4],
                            7, 41, (256, 400), False, 4),
    # BASELINE.json configs[0]'s model at FULL channel width: no decoder, the DensePose pooler over the four FPN levels at 14 x 14,
    # 15 coarse channels (configs/densepose_rcnn_R_50_FPN_s1x_legacy.yaml:7-13), reduced frame
    "full_r50_legacy_small": ("densepose_rcnn_R_50_FPN_s1x_legacy",
                              ["INPUT.MIN_SIZE_TEST", 256, "INPUT.MAX_SIZE_TEST", 448, "TEST.DETECTIONS_PER_IMAGE", 4],
                              8, 51, (256, 400), False, 4),
    # BASELINE.json configs[1] geometry: 800x1333 frame, R pinned to 8
    "full_r50_s1x_800x1333": ("densepose_rcnn_R_50_FPN_s1x", ["TEST.DETECTIONS_PER_IMAGE", 8], 0, 1234, (800, 1333), False, 8),
    # DeepLab head at its REAL geometry (densepose/config.py:177 POOLER_RESOLUTION 28; deeplab.py:33 rates 6 / 12 / 56):
    # live d = 6 and d = 12 taps on the 28x28 ROI map, d = 56 centre-tap only; full width = GroupNorm groups of 8 (ASPP,
    # 256 channels) and 16 (stacked convs, 512 channels) channels. BASELINE.json configs[3] at a reduced frame.
    "full_r50_dl_p28": ("densepose_rcnn_R_50_FPN_DL_s1x",
                        ["INPUT.MIN_SIZE_TEST", 192, "INPUT.MAX_SIZE_TEST", 320, "TEST.DETECTIONS_PER_IMAGE", 2],
                        5, 31, (192, 300), ("dp_head_out/8",), 4),
    # BASELINE.json configs[4] combination: R101 + DeepLab head (pool 28) on a 1080x1920 video frame (-> 749x1333);
    # tiny channel widths keep the CPU replay short - the geometry (padding 768x1344, live dilated taps) is the real one
    # BASELINE.json configs[4]'s combination at FULL channel width: R101 (res4 x 23) + DeepLab head at pool 28 (deeplab.py:64-144),
    # reduced frame: the only full-width meeting of those kernels with the reference
    "full_r101_dl_p28_small": ("densepose_rcnn_R_101_FPN_DL_s1x",
                               ["INPUT.MIN_SIZE_TEST", 192, "INPUT.MAX_SIZE_TEST", 320, "TEST.DETECTIONS_PER_IMAGE", 2],
                               9, 61, (192, 300), ("dp_head_out/8",), 4),
    "tiny_r101_dl_p28_video": ("densepose_rcnn_R_101_FPN_DL_s1x",
                               TINY_OPTS + ["INPUT.MIN_SIZE_TEST", 800, "INPUT.MAX_SIZE_TEST", 1333, "TEST.DETECTIONS_PER_IMAGE", 3],
                               6, 32, (1080, 1920), ("dp_head_out/1",), 4),
}


# The reference's OWN fp16 mode (run.py:26 / export.py:36-37: `predictor.half()`), run on the CPU: ATen's CPU kernels take half
# tensors (accumulating in float, rounding every layer's output to half). Recorded for cases that also have an fp32 golden, as
# `<case>__half.npz` (outputs + visualiser labels only): the distance fp16-reference <-> fp32-reference is the yardstick the
# engine's fp16 mode is held to (tests/test_gpu_e2e.py), and the engine must be about as close to this golden as to the fp32 one.
HALF_CASES = ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl", "full_r101_dl_p28_small"]
# ... and the reference in bfloat16 (`predictor.bfloat16()`: the dtype every throughput configuration of BASELINE.json names; the
# reference picks its dtype at run.py:20-29 / export.py:36-37 by calling .half() / .float() on the module - .bfloat16() is the same
# nn.Module call). `<case>__bf16.npz`: the yardstick for the engine's bf16 mode - how far does the REFERENCE ITSELF move from its fp32
# outputs when every weight and every layer output is rounded to bfloat16?
BF16_CASES = ["tiny_r50_s1x_a", "full_r50_s1x_small", "tiny_r50_dl", "full_r50_dl_p28", "full_r101_dl_p28_small", "full_r50_s1x_800x1333"]


def make_image(seed, hw):
    return np.random.default_rng(seed).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)


def _import_visualizer():
    if "cv2" not in sys.modules:  # cv2 is absent here; the drawing code that needs it is never called
        stub = types.ModuleType("cv2")
        stub.__getattr__ = lambda name: 0
        sys.modules["cv2"] = stub
    sys.path.insert(0, REFERENCE_ROOT)
    import visualizer  # noqa
    return visualizer


def run_half_case(name, mode="half"):
    cfg_name, opts, wseed, iseed, hw, stages, sub = CASES[name]
    cfg = get_config(cfg_name, opts)
    state = make_synthetic_state(cfg, wseed)
    pred = build_reference_predictor(cfg, state)
    pred = pred.half() if mode == "half" else pred.bfloat16()
    out = pred(torch.from_numpy(make_image(iseed, hw)))
    arrays = {}
    for k, v in out.items():
        a = v.float().numpy() if v.is_floating_point() else v.numpy()
        if k.startswith("pred_densepose") and sub > 1:
            a = a[:, :, ::sub, ::sub]
        arrays["out/" + k] = a
        arrays["dtype/" + k] = np.frombuffer(str(v.dtype).encode(), dtype=np.uint8)
    vis = _import_visualizer()
    results, _ = vis.DensePoseResultExtractor()({k: (v.float() if v.is_floating_point() else v) for k, v in out.items()})
    for i, r in enumerate(results):
        arrays["vis/labels_%d" % i] = r["labels"].numpy().astype(np.uint8)
    # ... and the same labels computed on the boxes of the FP32 golden's detections (nearest box within 1.5 px): the part-label agreement
    # of the reference's own low-precision run with its fp32 run - on identical pixel grids - is then a plain comparison with the fp32
    # golden's vis/labels_<i> (tests/test_oracle_golden.py, and the yardstick of the GPU tests)
    fp32_path = os.path.join(GOLDEN_DIR, name + ".npz")
    if os.path.exists(fp32_path):
        z32 = np.load(fp32_path)
        rb = z32["out/pred_boxes"]
        gb = out["pred_boxes"].float().numpy()
        match = np.full((len(rb),), -1, dtype=np.int32)
        for i in range(len(rb)):
            if len(gb) == 0:
                break
            d = np.abs(gb - rb[i]).max(axis=1)
            j = int(d.argmin())
            if d[j] < 1.5:
                match[i] = j
                inst = {k: out[k][j:j + 1].float() for k in out if k.startswith("pred_densepose")}
                inst["pred_boxes"] = torch.from_numpy(rb[i:i + 1].copy())
                (r,), _ = vis.DensePoseResultExtractor()(inst)
                arrays["vis/labels_on_fp32_box_%d" % i] = r["labels"].numpy().astype(np.uint8)
        arrays["match/fp32_to_this"] = match
    call = ".half()" if mode == "half" else ".bfloat16()"
    meta = dict(case=name + "__" + mode, config=cfg_name, opts=list(opts), weight_seed=wseed, image_seed=iseed, image_hw=list(hw),
                iuv_stride=sub, weights_sha256=state_checksum(state), torch=torch.__version__, mode="reference %s on CPU" % call,
                generator="oracle/make_goldens.py --%s (reference imported from /root/reference)" % mode)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLDEN_DIR, name + "__%s.npz" % mode)
    np.savez_compressed(path, **arrays)
    print("%-26s R=%d  %.2f MB (%s)" % (name, len(out["scores"]), os.path.getsize(path) / 1e6, mode), flush=True)


def run_case(name):
    cfg_name, opts, wseed, iseed, hw, stages, sub = CASES[name]
    cfg = get_config(cfg_name, opts)
    state = make_synthetic_state(cfg, wseed)
    pred = build_reference_predictor(cfg, state)
    model = pred.model
    cap = {}
    hooks = []
    subset = None
    if isinstance(stages, tuple):      # ("name/channel stride", ...): only these stage tensors, channel-subsampled
        subset = dict((k.split("/")[0], int(k.split("/")[1])) for k in stages)
        for nm, mod in (("dp_head_out", model.roi_heads.densepose_head), ("dp_pooled", model.roi_heads.densepose_pooler)):
            if nm in subset:
                hooks.append(mod.register_forward_hook((lambda n: (lambda m, i, o: cap.__setitem__(n, o)))(nm)))
        stages = False
    if stages:
        def hook(nm):
            def fn(mod, inp, out):
                cap[nm] = out
            return fn
        hooks.append(model.backbone.register_forward_hook(hook("features")))
        hooks.append(model.proposal_generator.register_forward_hook(hook("proposals")))
        hooks.append(model.proposal_generator.rpn_head.register_forward_hook(hook("rpn_head")))
        hooks.append(model.roi_heads.box_pooler.register_forward_hook(hook("box_pooled")))
        hooks.append(model.roi_heads.box_predictor.register_forward_hook(hook("box_predictor")))
        if model.roi_heads.decoder is not None:
            hooks.append(model.roi_heads.decoder.register_forward_hook(hook("decoder_out")))
        hooks.append(model.roi_heads.densepose_pooler.register_forward_hook(hook("dp_pooled")))
        hooks.append(model.roi_heads.densepose_head.register_forward_hook(hook("dp_head_out")))
    img = torch.from_numpy(make_image(iseed, hw))
    out = pred(img)
    for h in hooks:
        h.remove()
    arrays = {}
    for k, v in out.items():
        a = v.numpy()
        if k.startswith("pred_densepose") and sub > 1:
            a = a[:, :, ::sub, ::sub]
        arrays["out/" + k] = a
    # part-index argmax + uv as the visualiser computes them
    vis = _import_visualizer()
    results, _ = vis.DensePoseResultExtractor()(out)
    for i, r in enumerate(results):
        arrays["vis/labels_%d" % i] = r["labels"].numpy().astype(np.uint8)
        arrays["vis/uv_%d" % i] = r["uv"].numpy()
    if subset:
        for nm, cs in subset.items():
            arrays["stage/" + nm] = cap[nm].numpy()[:, ::cs]
    if stages:
        for k, v in cap["features"].items():
            arrays["stage/" + k] = v.numpy()
        props = cap["proposals"][0][0]
        arrays["stage/proposal_boxes"] = props["proposal_boxes"].numpy()
        arrays["stage/objectness_logits"] = props["objectness_logits"].numpy()
        logits, deltas = cap["rpn_head"]
        for i, (l, d) in enumerate(zip(logits, deltas)):
            arrays["stage/rpn_logits_%d" % i] = l.numpy()
            arrays["stage/rpn_deltas_%d" % i] = d.numpy()
        arrays["stage/box_pooled"] = cap["box_pooled"].numpy()
        arrays["stage/box_logits"] = cap["box_predictor"][0].numpy()
        arrays["stage/box_deltas"] = cap["box_predictor"][1].numpy()
        if "decoder_out" in cap:
            arrays["stage/decoder_out"] = cap["decoder_out"].numpy()
        arrays["stage/dp_pooled"] = cap["dp_pooled"].numpy()
        arrays["stage/dp_head_out"] = cap["dp_head_out"].numpy()
    meta = dict(case=name, config=cfg_name, opts=list(opts), weight_seed=wseed, image_seed=iseed, image_hw=list(hw),
                iuv_stride=sub, stage_channel_stride=(subset or {}), weights_sha256=state_checksum(state), torch=torch.__version__,
                generator="oracle/make_goldens.py (reference imported from /root/reference)")
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-26s R=%d  %.2f MB" % (name, len(out["scores"]), os.path.getsize(path) / 1e6), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--half", action="store_true", help="record the <case>__half.npz fixtures (reference .half() on CPU) instead")
    ap.add_argument("--bf16", action="store_true", help="record the <case>__bf16.npz fixtures (reference .bfloat16() on CPU) instead")
    args = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 1)
    if args.half or args.bf16:
        for name in (HALF_CASES if args.half else BF16_CASES):
            if not args.only or args.only == name:
                run_half_case(name, "half" if args.half else "bf16")
        return
    for name in CASES:
        if args.only and args.only != name:
            continue
        run_case(name)


if __name__ == "__main__":
    main()

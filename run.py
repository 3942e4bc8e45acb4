#!/usr/bin/env python3
"""Counterpart of the reference's run.py (image in -> DensePose IUV out), on the MI355X engine.

    python run.py <config name | yaml> <weights.pkl | synthetic[:seed]> <image.(png|jpg|npy)> [--out result.npz] [--fp32]

The reference CLI (run.py:18-37) loads a TorchScript file, reads the frame with cv2 and draws an overlay; here the model
is built from the config + Detectron2-zoo .pkl, the frame is read with PIL/numpy (cv2 is not a dependency) and the IUV
array the reference would colour-map is written instead of a drawing."""
import argparse

import numpy as np
import torch


def load_image(path):
    if path.endswith(".npy"):
        a = np.load(path)
    else:
        from PIL import Image
        a = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]  # RGB -> BGR like cv2.imread
    assert a.ndim == 3 and a.shape[2] == 3 and a.dtype == np.uint8, a.shape
    return np.ascontiguousarray(a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("weights")
    ap.add_argument("input")
    ap.add_argument("--out", default="densepose_out.npz")
    ap.add_argument("--fp32", action="store_true", help="parity mode (exact fp32 MFMA) instead of bf16")
    ap.add_argument("--fp16", action="store_true", help="IEEE half operands (the reference's export.py --fp16 / run.py default on GPU)")
    ap.add_argument("--min_score", type=float, default=0.3)
    args = ap.parse_args()
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from densepose_torchscript_amd.visualizer import extract_iuv, iuv_image
    cfg = get_config(args.config, ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", args.min_score])
    weights = args.weights
    if weights.startswith("synthetic"):
        weights = make_synthetic_state(cfg, int(weights.split(":")[1]) if ":" in weights else 0)
    predictor = DensePosePredictor(cfg, weights, dtype="fp32" if args.fp32 else "fp16" if args.fp16 else "bf16")
    img = load_image(args.input)
    outputs = predictor(torch.from_numpy(img))
    results, xywh = extract_iuv(outputs)
    iuv = iuv_image(results, xywh, img.shape[0], img.shape[1])
    np.savez_compressed(args.out, iuv=iuv, pred_boxes=outputs["pred_boxes"].cpu().numpy(), scores=outputs["scores"].cpu().numpy())
    print("%d detections -> %s" % (len(results), args.out))


if __name__ == "__main__":
    main()

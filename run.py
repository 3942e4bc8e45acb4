#!/usr/bin/env python3
"""Counterpart of the reference's run.py (image in -> DensePose IUV out), on the MI355X engine.

    python run.py <config name | yaml> <weights.pkl | synthetic[:seed]> <image.(png|jpg|npy) | frames.npy> [--out result.npz] [--fp32]

A 4-D ``.npy`` ([T, H, W, 3] uint8 BGR) is treated like the reference's video loop (run.py:42-57): the frames go through
``predict_batch`` in batches of ``--batch`` on two pipeline lanes and the output holds one IUV array per frame.

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
    assert a.ndim in (3, 4) and a.shape[-1] == 3 and a.dtype == np.uint8, a.shape
    return np.ascontiguousarray(a)


def run_frames(predictor, frames, batch):
    """The reference's per-frame video loop (run.py:42-57) as batches of equal-size frames on two pipeline lanes."""
    from densepose_torchscript_amd.visualizer import extract_iuv, iuv_image
    predictor.pipeline_depth = 2
    pending, iuvs, n_det = [], [], 0

    def drain(keep):
        nonlocal n_det
        while len(pending) > keep:
            outs, done = pending.pop(0)
            predictor.wait(done)    # this stream waits for the batch it consumes only; the newer batch keeps its lane busy
            for o in outs:
                results, xywh = extract_iuv(o)
                h, w = [int(v) for v in o["image_size"]]
                iuvs.append(iuv_image(results, xywh, h, w))
                n_det += len(results)

    for i in range(0, len(frames), batch):
        chunk = [torch.from_numpy(f).to(predictor.device, non_blocking=True) for f in frames[i:i + batch]]
        pending.append((predictor.predict_batch(chunk), predictor.completion()))
        drain(1)                    # post-process batch i-1 while batch i runs
    drain(0)
    return np.stack(iuvs), n_det


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("weights")
    ap.add_argument("input")
    ap.add_argument("--out", default="densepose_out.npz")
    ap.add_argument("--fp32", action="store_true", help="parity mode (exact fp32 MFMA) instead of bf16")
    ap.add_argument("--fp16", action="store_true", help="IEEE half operands (the reference's export.py --fp16 / run.py default on GPU)")
    ap.add_argument("--min_score", type=float, default=0.3)
    ap.add_argument("--batch", type=int, default=8, help="frames per batch for a [T,H,W,3] .npy input")
    ap.add_argument("--opts", nargs="*", default=[], metavar="KEY VALUE", help="config overrides like the reference's `opts` (detectron2 yacs keys)")
    args = ap.parse_args()
    from densepose_torchscript_amd import get_config, make_synthetic_state
    from densepose_torchscript_amd.predictor import DensePosePredictor
    from densepose_torchscript_amd.visualizer import extract_iuv, iuv_image
    extra = [int(v) if v.lstrip("-").isdigit() else v for v in args.opts]     # extra "KEY value" overrides (tests: tiny widths)
    cfg = get_config(args.config, extra + ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", args.min_score])
    weights = args.weights
    if weights.startswith("synthetic"):
        weights = make_synthetic_state(cfg, int(weights.split(":")[1]) if ":" in weights else 0)
    predictor = DensePosePredictor(cfg, weights, dtype="fp32" if args.fp32 else "fp16" if args.fp16 else "bf16", resize="device",
                                   use_graphs=True)
    img = load_image(args.input)
    if img.ndim == 4:
        iuv, n_det = run_frames(predictor, img, args.batch)
        np.savez_compressed(args.out, iuv=iuv)
        print("%d frames, %d detections -> %s" % (len(img), n_det, args.out))
        return
    outputs = predictor(torch.from_numpy(img))
    results, xywh = extract_iuv(outputs)
    iuv = iuv_image(results, xywh, img.shape[0], img.shape[1])
    np.savez_compressed(args.out, iuv=iuv, pred_boxes=outputs["pred_boxes"].cpu().numpy(), scores=outputs["scores"].cpu().numpy())
    print("%d detections -> %s" % (len(results), args.out))


if __name__ == "__main__":
    main()

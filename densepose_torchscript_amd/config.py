"""Model hyper-parameters of the DensePose R-CNN inference path.

A frozen dataclass per model variant replaces the reference's yacs tree
(/root/reference/detectron2/config.py:96-713, densepose/config.py:158-277,
configs/Base-DensePose-RCNN-FPN.yaml, configs/densepose_rcnn_R_{50,101}_FPN[_DL]_s1x[_legacy].yaml).
Only the keys that reach the inference path are kept; ``from_yaml`` / ``with_overrides``
accept the reference's own dotted key names so a reference yaml (with ``_BASE_``
inheritance) or an ``export.py``-style override list can be used unchanged.
"""
import dataclasses
import os
from ast import literal_eval
from dataclasses import dataclass
from typing import Tuple

import yaml


@dataclass(frozen=True)
class ModelConfig:
    name: str = "densepose_rcnn_R_50_FPN_s1x"
    # INPUT.* (config.py:136-156)
    min_size: int = 800
    max_size: int = 1333
    input_format: str = "BGR"
    pixel_mean: Tuple[float, float, float] = (103.530, 116.280, 123.675)
    pixel_std: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    # MODEL.RESNETS.* (config.py:546-569)
    depth: int = 50
    stem_out: int = 64
    res2_out: int = 256
    width_per_group: int = 64
    # MODEL.FPN.OUT_CHANNELS
    fpn_out: int = 256
    # MODEL.ANCHOR_GENERATOR.*
    anchor_sizes: Tuple[float, ...] = (32.0, 64.0, 128.0, 256.0, 512.0)
    anchor_ratios: Tuple[float, ...] = (0.5, 1.0, 2.0)
    # MODEL.RPN.*
    rpn_pre_topk: int = 1000
    rpn_post_topk: int = 1000
    rpn_nms_thresh: float = 0.7
    # MODEL.ROI_BOX_HEAD.* / MODEL.ROI_HEADS.* / TEST.*
    box_pool: int = 7
    box_sampling: int = 2
    box_fc_dim: int = 1024
    box_num_fc: int = 2
    bbox_reg_weights: Tuple[float, float, float, float] = (10.0, 10.0, 5.0, 5.0)
    score_thresh: float = 0.3  # export.py:15,23-24 (min_score), NOT the yaml default 0.05
    nms_thresh: float = 0.5
    dets_per_image: int = 100
    # MODEL.ROI_DENSEPOSE_HEAD.*
    dp_head: str = "DensePoseV1ConvXHead"  # or "DensePoseDeepLabHead"
    dp_decoder_on: bool = True
    dp_decoder_dims: int = 256
    dp_decoder_classes: int = 256
    dp_pool: int = 28
    dp_sampling: int = 2
    dp_head_dim: int = 512
    dp_num_convs: int = 8
    dp_coarse_ch: int = 2
    dp_patches: int = 24

    # ---- derived ----
    @property
    def blocks_per_stage(self):
        return {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[self.depth]

    @property
    def is_deeplab(self):
        return self.dp_head == "DensePoseDeepLabHead"

    @property
    def heatmap(self):
        # deconv x2 then bilinear x2 (chart.py:45-60,72-74)
        return self.dp_pool * 4

    def with_overrides(self, opts):
        """opts: flat list [KEY, value, KEY, value ...] with the reference's dotted keys."""
        kw = {}
        assert len(opts) % 2 == 0
        for k, v in zip(opts[0::2], opts[1::2]):
            if isinstance(v, str):
                try:
                    v = literal_eval(v)
                except (ValueError, SyntaxError):
                    pass
            if k in _FIXED_KEYS:
                # semantics the kernels hard-wire: accepted only with the value they implement, never silently ignored
                allowed = _FIXED_KEYS[k]
                if v not in allowed:
                    raise ValueError("%s = %r is not supported by the MI355X path (implemented: %s)"
                                     % (k, v, " / ".join(repr(a) for a in allowed)))
                continue
            if k not in _KEYMAP:
                if k in _IGNORED_KEYS or k.split(".")[0] in ("SOLVER", "DATASETS", "DATALOADER"):
                    continue
                raise KeyError("unsupported config key for the inference path: " + k)
            field, conv = _KEYMAP[k]
            kw[field] = conv(v)
        if kw.get("dp_head", self.dp_head) not in ("DensePoseV1ConvXHead", "DensePoseDeepLabHead"):
            raise ValueError("MODEL.ROI_DENSEPOSE_HEAD.NAME = %r is not supported (DensePoseV1ConvXHead / DensePoseDeepLabHead)" % kw["dp_head"])
        if kw.get("input_format", self.input_format) not in ("BGR", "RGB"):
            raise ValueError("INPUT.FORMAT = %r is not supported (BGR / RGB)" % kw["input_format"])
        return dataclasses.replace(self, **kw)

    @staticmethod
    def from_yaml(path, opts=()):
        tree = _load_yaml_with_base(path)
        flat = []
        _flatten(tree, "", flat)
        name = os.path.splitext(os.path.basename(path))[0]
        cfg = ModelConfig(name=name).with_overrides(flat)
        return cfg.with_overrides(list(opts))


def _sizes(v):
    return tuple(float(s[0]) for s in v)


_KEYMAP = {
    "INPUT.MIN_SIZE_TEST": ("min_size", int),
    "INPUT.MAX_SIZE_TEST": ("max_size", int),
    "INPUT.FORMAT": ("input_format", str),
    "MODEL.PIXEL_MEAN": ("pixel_mean", lambda v: tuple(float(x) for x in v)),
    "MODEL.PIXEL_STD": ("pixel_std", lambda v: tuple(float(x) for x in v)),
    "MODEL.RESNETS.DEPTH": ("depth", int),
    "MODEL.RESNETS.STEM_OUT_CHANNELS": ("stem_out", int),
    "MODEL.RESNETS.RES2_OUT_CHANNELS": ("res2_out", int),
    "MODEL.RESNETS.WIDTH_PER_GROUP": ("width_per_group", int),
    "MODEL.FPN.OUT_CHANNELS": ("fpn_out", int),
    "MODEL.ANCHOR_GENERATOR.SIZES": ("anchor_sizes", _sizes),
    "MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS": ("anchor_ratios", lambda v: tuple(float(x) for x in v[0])),
    "MODEL.RPN.PRE_NMS_TOPK_TEST": ("rpn_pre_topk", int),
    "MODEL.RPN.POST_NMS_TOPK_TEST": ("rpn_post_topk", int),
    "MODEL.RPN.NMS_THRESH": ("rpn_nms_thresh", float),
    "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION": ("box_pool", int),
    "MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO": ("box_sampling", int),
    "MODEL.ROI_BOX_HEAD.FC_DIM": ("box_fc_dim", int),
    "MODEL.ROI_BOX_HEAD.NUM_FC": ("box_num_fc", int),
    "MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS": ("bbox_reg_weights", lambda v: tuple(float(x) for x in v)),
    "MODEL.ROI_HEADS.SCORE_THRESH_TEST": ("score_thresh", float),
    "MODEL.ROI_HEADS.NMS_THRESH_TEST": ("nms_thresh", float),
    "TEST.DETECTIONS_PER_IMAGE": ("dets_per_image", int),
    "MODEL.ROI_DENSEPOSE_HEAD.NAME": ("dp_head", str),
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_ON": ("dp_decoder_on", bool),
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_CONV_DIMS": ("dp_decoder_dims", int),
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_NUM_CLASSES": ("dp_decoder_classes", int),
    "MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION": ("dp_pool", int),
    "MODEL.ROI_DENSEPOSE_HEAD.POOLER_SAMPLING_RATIO": ("dp_sampling", int),
    "MODEL.ROI_DENSEPOSE_HEAD.CONV_HEAD_DIM": ("dp_head_dim", int),
    "MODEL.ROI_DENSEPOSE_HEAD.NUM_STACKED_CONVS": ("dp_num_convs", int),
    "MODEL.ROI_DENSEPOSE_HEAD.NUM_COARSE_SEGM_CHANNELS": ("dp_coarse_ch", int),
    "MODEL.ROI_DENSEPOSE_HEAD.NUM_PATCHES": ("dp_patches", int),
}

# Keys whose semantics are hard-wired in the kernels / the engine: a yaml or override may state them, but only with the value
# that is implemented - anything else raises instead of being computed with the wrong semantics (e.g. the reference's own
# default POOLER_TYPE is ROIAlignV2 = aligned=True, densepose/config.py:176; the BASELINE yamls set ROIAlign, Base-...yaml:33,36).
_FIXED_KEYS = {
    "MODEL.META_ARCHITECTURE": ("GeneralizedRCNN",),
    "MODEL.BACKBONE.NAME": ("build_resnet_fpn_backbone",),
    "MODEL.RESNETS.OUT_FEATURES": (["res2", "res3", "res4", "res5"], ("res2", "res3", "res4", "res5")),
    "MODEL.RESNETS.STRIDE_IN_1X1": (True,),
    "MODEL.RESNETS.NUM_GROUPS": (1,),
    "MODEL.RESNETS.RES5_DILATION": (1,),
    "MODEL.RESNETS.NORM": ("FrozenBN",),
    "MODEL.FPN.IN_FEATURES": (["res2", "res3", "res4", "res5"], ("res2", "res3", "res4", "res5")),
    "MODEL.FPN.NORM": ("",),
    "MODEL.FPN.FUSE_TYPE": ("sum",),
    "MODEL.RPN.IN_FEATURES": (["p2", "p3", "p4", "p5", "p6"], ("p2", "p3", "p4", "p5", "p6")),
    "MODEL.RPN.HEAD_NAME": ("StandardRPNHead",),
    "MODEL.PROPOSAL_GENERATOR.NAME": ("RPN",),
    "MODEL.PROPOSAL_GENERATOR.MIN_SIZE": (0,),
    "MODEL.ANCHOR_GENERATOR.OFFSET": (0.0, 0),
    "MODEL.DENSEPOSE_ON": (True,),
    "MODEL.MASK_ON": (False,),
    "MODEL.KEYPOINT_ON": (False,),
    "MODEL.ROI_HEADS.NAME": ("DensePoseROIHeads",),
    "MODEL.ROI_HEADS.IN_FEATURES": (["p2", "p3", "p4", "p5"], ("p2", "p3", "p4", "p5")),
    "MODEL.ROI_HEADS.NUM_CLASSES": (1,),
    "MODEL.ROI_BOX_HEAD.NAME": ("FastRCNNConvFCHead",),
    "MODEL.ROI_BOX_HEAD.POOLER_TYPE": ("ROIAlign",),           # aligned=False (poolers.py:149-155)
    "MODEL.ROI_BOX_HEAD.NUM_CONV": (0,),
    "MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG": (False,),
    "MODEL.ROI_DENSEPOSE_HEAD.POOLER_TYPE": ("ROIAlign",),
    "MODEL.ROI_DENSEPOSE_HEAD.DECONV_KERNEL": (4,),            # chart.py:45-59: the four sub-pixel 2x2 convolutions
    "MODEL.ROI_DENSEPOSE_HEAD.UP_SCALE": (2,),                 # chart.py:72-74: bilinear x2
    "MODEL.ROI_DENSEPOSE_HEAD.CONV_HEAD_KERNEL": (3,),
    "MODEL.ROI_DENSEPOSE_HEAD.PREDICTOR_NAME": ("DensePoseChartWithConfidencePredictor", "DensePoseChartPredictor"),
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_COMMON_STRIDE": (4,),
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_NORM": ("",),
    "MODEL.ROI_DENSEPOSE_HEAD.UV_CONFIDENCE.ENABLED": (False,),   # confidence heads add output keys: out of scope
    "MODEL.ROI_DENSEPOSE_HEAD.SEGM_CONFIDENCE.ENABLED": (False,),
    "MODEL.ROI_DENSEPOSE_HEAD.DEEPLAB.NORM": ("GN",),
    "MODEL.ROI_DENSEPOSE_HEAD.DEEPLAB.NONLOCAL_ON": (0, False),
}

# keys that appear in the reference yamls but do not change the inference path (training-only or bookkeeping)
_IGNORED_KEYS = {
    "VERSION", "_BASE_", "MODEL.RPN.PRE_NMS_TOPK_TRAIN", "MODEL.RPN.POST_NMS_TOPK_TRAIN", "MODEL.WEIGHTS",
    "INPUT.MIN_SIZE_TRAIN", "MODEL.ROI_DENSEPOSE_HEAD.HEATMAP_SIZE",
    "MODEL.ROI_DENSEPOSE_HEAD.INDEX_WEIGHTS", "MODEL.ROI_DENSEPOSE_HEAD.PART_WEIGHTS",
    "MODEL.ROI_DENSEPOSE_HEAD.POINT_REGRESSION_WEIGHTS", "MODEL.DEVICE",
    "MODEL.ROI_DENSEPOSE_HEAD.UV_CONFIDENCE.EPSILON", "MODEL.ROI_DENSEPOSE_HEAD.UV_CONFIDENCE.TYPE",
    "MODEL.ROI_DENSEPOSE_HEAD.SEGM_CONFIDENCE.EPSILON", "MODEL.ROI_DENSEPOSE_HEAD.COARSE_SEGM_TRAINED_BY_MASKS",
}


def _load_yaml_with_base(path):
    with open(path, "r") as f:
        cfg = yaml.safe_load(f) or {}
    base = cfg.pop("_BASE_", None)
    if base is None:
        return cfg
    if not os.path.isabs(base):
        base = os.path.join(os.path.dirname(path), base)
    merged = _load_yaml_with_base(base)

    def merge(a, b):
        for k, v in a.items():
            if isinstance(v, dict) and isinstance(b.get(k), dict):
                merge(v, b[k])
            else:
                b[k] = v

    merge(cfg, merged)
    return merged


def _flatten(tree, prefix, out):
    for k, v in tree.items():
        key = prefix + k
        if isinstance(v, dict):
            _flatten(v, key + ".", out)
        else:
            out.extend([key, v])


# The five BASELINE.json configurations (+ the R101 legacy zoo model).
_VARIANTS = {
    "densepose_rcnn_R_50_FPN_s1x": {},
    "densepose_rcnn_R_101_FPN_s1x": dict(depth=101),
    "densepose_rcnn_R_50_FPN_s1x_legacy": dict(dp_coarse_ch=15, dp_pool=14, dp_decoder_on=False),
    "densepose_rcnn_R_101_FPN_s1x_legacy": dict(depth=101, dp_coarse_ch=15, dp_pool=14, dp_decoder_on=False),
    "densepose_rcnn_R_50_FPN_DL_s1x": dict(dp_head="DensePoseDeepLabHead"),
    "densepose_rcnn_R_101_FPN_DL_s1x": dict(depth=101, dp_head="DensePoseDeepLabHead"),
}

# Tiny-width variants of the same code paths: identical topology, ~1 M parameters.
TINY_OPTS = [
    "MODEL.RESNETS.STEM_OUT_CHANNELS", 16, "MODEL.RESNETS.RES2_OUT_CHANNELS", 32,
    "MODEL.RESNETS.WIDTH_PER_GROUP", 8, "MODEL.FPN.OUT_CHANNELS", 32,
    "MODEL.ROI_BOX_HEAD.FC_DIM", 64, "MODEL.ROI_DENSEPOSE_HEAD.CONV_HEAD_DIM", 64,
    "MODEL.ROI_DENSEPOSE_HEAD.DECODER_CONV_DIMS", 32, "MODEL.ROI_DENSEPOSE_HEAD.DECODER_NUM_CLASSES", 32,
    "INPUT.MIN_SIZE_TEST", 128, "INPUT.MAX_SIZE_TEST", 213,
    "MODEL.RPN.PRE_NMS_TOPK_TEST", 200, "MODEL.RPN.POST_NMS_TOPK_TEST", 100,
    "TEST.DETECTIONS_PER_IMAGE", 4,
]


def get_config(name, opts=()):
    if name.endswith(".yaml"):
        return ModelConfig.from_yaml(name, opts)
    base = name
    if base not in _VARIANTS:
        raise KeyError("unknown model variant %r (known: %s)" % (name, ", ".join(sorted(_VARIANTS))))
    cfg = ModelConfig(name=base, **_VARIANTS[base])
    return cfg.with_overrides(list(opts))

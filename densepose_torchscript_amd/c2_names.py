"""Caffe2 / Detectron1 blob names -> the canonical (Detectron2) parameter names of ``weights.param_shapes``.

The reference loads such checkpoints through a chain of string rewrites followed by a longest-suffix match against the
model's own key list (/root/reference/detectron2/checkpoint/c2_model_loading.py:66-204, :207-299,
detection_checkpoint.py:57-64, :95-103). Here the same mapping is written as ONE ordered rule table per name family of
the DensePose R-CNN FPN models (trunk, FPN, RPN, box head, DensePose head); a blob that no rule covers is reported,
never guessed. Value fix-ups of the reference are kept: Detectron1 puts the background class first
(``cls_score`` rows are rotated so it comes last, the first 4 ``bbox_pred`` rows are dropped), and its frozen BN is a bare
affine pair ``*_bn_s / *_bn_b`` - the running statistics take the values a fresh FrozenBatchNorm2d has
(mean 0, var 1 - eps, so that the folded scale is exactly ``s``; batch_norm.py:38).
"""
import re
from collections import OrderedDict

import numpy as np

_SUFFIX = {"w": "weight", "b": "bias"}

# (compiled pattern on the blob name, replacement template for the canonical key WITHOUT the trailing ".weight"/".bias")
_RULES = [
    # --- ResNet trunk ---------------------------------------------------------------------------------------------
    (r"^conv1$", "backbone.bottom_up.stem.conv1"),
    (r"^res_conv1_bn$", "backbone.bottom_up.stem.conv1.norm"),
    (r"^res(\d)_(\d+)_branch1$", r"backbone.bottom_up.res\1.\2.shortcut"),
    (r"^res(\d)_(\d+)_branch1_bn$", r"backbone.bottom_up.res\1.\2.shortcut.norm"),
    (r"^res(\d)_(\d+)_branch2a$", r"backbone.bottom_up.res\1.\2.conv1"),
    (r"^res(\d)_(\d+)_branch2a_bn$", r"backbone.bottom_up.res\1.\2.conv1.norm"),
    (r"^res(\d)_(\d+)_branch2b$", r"backbone.bottom_up.res\1.\2.conv2"),
    (r"^res(\d)_(\d+)_branch2b_bn$", r"backbone.bottom_up.res\1.\2.conv2.norm"),
    (r"^res(\d)_(\d+)_branch2c$", r"backbone.bottom_up.res\1.\2.conv3"),
    (r"^res(\d)_(\d+)_branch2c_bn$", r"backbone.bottom_up.res\1.\2.conv3.norm"),
    # --- FPN: "fpn_inner_res5_2_sum" (top lateral), "fpn_inner_res4_5_sum_lateral", "fpn_res4_5_sum" (3x3 output) --------
    (r"^fpn_inner_res(\d)_\d+_sum(?:_lateral)?$", r"backbone.fpn_lateral\1"),
    (r"^fpn_res(\d)_\d+_sum$", r"backbone.fpn_output\1"),
    # --- RPN (defined on level 2 and shared, hence "fpn2") ----------------------------------------------------------------
    (r"^conv_rpn(?:_fpn2)?$", "proposal_generator.rpn_head.conv"),
    (r"^rpn_cls_logits(?:_fpn2)?$", "proposal_generator.rpn_head.objectness_logits"),
    (r"^rpn_bbox_pred(?:_fpn2)?$", "proposal_generator.rpn_head.anchor_deltas"),
    # --- box head -----------------------------------------------------------------------------------------------------
    (r"^fc6$", "roi_heads.box_head.fc1"),
    (r"^fc7$", "roi_heads.box_head.fc2"),
    (r"^cls_score$", "roi_heads.box_predictor.cls_score"),
    (r"^bbox_pred$", "roi_heads.box_predictor.bbox_pred"),
    # --- DensePose head / predictor ------------------------------------------------------------------------------------
    (r"^body_conv_fcn(\d+)$", r"roi_heads.densepose_head.body_conv_fcn\1"),
    (r"^AnnIndex_lowres$", "roi_heads.densepose_predictor.ann_index_lowres"),
    (r"^Index_UV_lowres$", "roi_heads.densepose_predictor.index_uv_lowres"),
    (r"^U_lowres$", "roi_heads.densepose_predictor.u_lowres"),
    (r"^V_lowres$", "roi_heads.densepose_predictor.v_lowres"),
]
_RULES = [(re.compile(p), t) for p, t in _RULES]
_BN_EPS = 1e-5


def canonical_c2_key(blob):
    """'res2_0_branch2a_bn_s' -> 'backbone.bottom_up.res2.0.conv1.norm.weight' ; None if no rule covers the blob."""
    m = re.match(r"^(.*)_(w|b|s)$", blob)
    if not m:
        return None
    stem, kind = m.group(1), m.group(2)
    if kind == "s":           # affine scale of a frozen BN ("*_bn_s"); its bias is "*_bn_b"
        kind = "w"
    for rx, tmpl in _RULES:
        mm = rx.match(stem)
        if mm:
            return mm.expand(tmpl) + "." + _SUFFIX[kind]
    return None


def convert_caffe2_blobs(blobs, want_shapes):
    """blobs: Caffe2 name -> ndarray (momentum blobs already dropped). want_shapes: ``param_shapes(cfg)``.
    Returns an OrderedDict canonical key -> float32 ndarray holding every key of ``want_shapes`` the checkpoint determines."""
    out = OrderedDict()
    unknown = []
    for name in sorted(blobs):
        key = canonical_c2_key(name)
        if key is None:
            unknown.append(name)
            continue
        v = np.asarray(blobs[name], dtype=np.float32)
        if key.startswith("roi_heads.box_predictor.cls_score."):
            v = np.concatenate([v[1:], v[:1]])      # background class: index 0 in Detectron1, last in Detectron2
        elif key.startswith("roi_heads.box_predictor.bbox_pred."):
            v = v[4:]                                 # no box regression for the background class
        if key in out:
            raise ValueError("Caffe2 blobs %r and an earlier one both map to %s" % (name, key))
        out[key] = v
    for key in list(out):      # frozen BN affine pairs carry no running statistics
        if key.endswith(".norm.weight"):
            base = key[: -len("weight")]
            c = out[key].shape[0]
            out.setdefault(base + "running_mean", np.zeros((c,), np.float32))
            out.setdefault(base + "running_var", np.full((c,), 1.0 - _BN_EPS, np.float32))
    if unknown:
        raise ValueError("Caffe2 blobs without a conversion rule (DensePose R-CNN FPN families only): %s" % unknown[:8])
    extra = [k for k in out if k not in want_shapes]
    if extra:
        raise ValueError("converted keys that this model variant does not have: %s" % extra[:8])
    return out

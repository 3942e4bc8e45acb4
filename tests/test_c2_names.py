"""Caffe2 / Detectron1 blob names -> canonical keys (c2_model_loading.py:66-204)."""
import pickle

import numpy as np
import pytest

from densepose_torchscript_amd import TINY_OPTS, get_config
from densepose_torchscript_amd.c2_names import canonical_c2_key, convert_caffe2_blobs
from densepose_torchscript_amd.weights import check_state, load_checkpoint, make_synthetic_state, param_shapes


def _to_caffe2_blobs(cfg, state):
    """Write a canonical state the way Detectron1 named and laid out its blobs (the inverse of the converter, for tests)."""
    blobs = {}
    for k, v in state.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            continue                                  # Detectron1 has only the folded affine pair
        stem, kind = k.rsplit(".", 1)
        sfx = {"weight": "w", "bias": "b"}[kind]
        m = None
        import re
        if stem == "backbone.bottom_up.stem.conv1":
            name = "conv1"
        elif stem == "backbone.bottom_up.stem.conv1.norm":
            name, sfx = "res_conv1_bn", {"w": "s", "b": "b"}[sfx]
        elif (m := re.match(r"backbone\.bottom_up\.res(\d)\.(\d+)\.(shortcut|conv1|conv2|conv3)(\.norm)?$", stem)):
            br = {"shortcut": "branch1", "conv1": "branch2a", "conv2": "branch2b", "conv3": "branch2c"}[m.group(3)]
            name = "res%s_%s_%s" % (m.group(1), m.group(2), br)
            if m.group(4):
                name, sfx = name + "_bn", {"w": "s", "b": "b"}[sfx]
        elif (m := re.match(r"backbone\.fpn_lateral(\d)$", stem)):
            last = {2: 2, 3: 3, 4: cfg.blocks_per_stage[2] - 1, 5: 2}[int(m.group(1))]
            name = "fpn_inner_res%s_%d_sum%s" % (m.group(1), last, "" if m.group(1) == "5" else "_lateral")
        elif (m := re.match(r"backbone\.fpn_output(\d)$", stem)):
            last = {2: 2, 3: 3, 4: cfg.blocks_per_stage[2] - 1, 5: 2}[int(m.group(1))]
            name = "fpn_res%s_%d_sum" % (m.group(1), last)
        elif stem.startswith("proposal_generator.rpn_head."):
            name = {"conv": "conv_rpn_fpn2", "objectness_logits": "rpn_cls_logits_fpn2", "anchor_deltas": "rpn_bbox_pred_fpn2"}[stem.split(".")[-1]]
        elif stem.startswith("roi_heads.box_head."):
            name = {"fc1": "fc6", "fc2": "fc7"}[stem.split(".")[-1]]
        elif stem == "roi_heads.box_predictor.cls_score":
            name, v = "cls_score", np.concatenate([v[-1:], v[:-1]])            # background first
        elif stem == "roi_heads.box_predictor.bbox_pred":
            name, v = "bbox_pred", np.concatenate([np.full((4,) + v.shape[1:], 7.0, np.float32), v])  # 4 background rows
        elif (m := re.match(r"roi_heads\.densepose_head\.(body_conv_fcn\d+)$", stem)):
            name = m.group(1)
        elif stem.startswith("roi_heads.densepose_predictor."):
            name = {"ann_index_lowres": "AnnIndex_lowres", "index_uv_lowres": "Index_UV_lowres", "u_lowres": "U_lowres",
                    "v_lowres": "V_lowres"}[stem.split(".")[-1]]
        else:
            raise AssertionError(stem)
        blobs[name + "_" + sfx] = v
        if sfx == "w" and not name.endswith("_bn"):
            blobs[name + "_w_momentum"] = np.zeros_like(v)     # training leftovers the loader must drop
    return blobs


def test_key_rules():
    assert canonical_c2_key("res2_0_branch2a_w") == "backbone.bottom_up.res2.0.conv1.weight"
    assert canonical_c2_key("res4_22_branch2c_bn_s") == "backbone.bottom_up.res4.22.conv3.norm.weight"
    assert canonical_c2_key("res3_0_branch1_bn_b") == "backbone.bottom_up.res3.0.shortcut.norm.bias"
    assert canonical_c2_key("res_conv1_bn_s") == "backbone.bottom_up.stem.conv1.norm.weight"
    assert canonical_c2_key("conv1_w") == "backbone.bottom_up.stem.conv1.weight"
    assert canonical_c2_key("fpn_inner_res5_2_sum_w") == "backbone.fpn_lateral5.weight"
    assert canonical_c2_key("fpn_inner_res4_5_sum_lateral_b") == "backbone.fpn_lateral4.bias"
    assert canonical_c2_key("fpn_res3_3_sum_w") == "backbone.fpn_output3.weight"
    assert canonical_c2_key("conv_rpn_fpn2_w") == "proposal_generator.rpn_head.conv.weight"
    assert canonical_c2_key("rpn_bbox_pred_fpn2_b") == "proposal_generator.rpn_head.anchor_deltas.bias"
    assert canonical_c2_key("fc6_w") == "roi_heads.box_head.fc1.weight"
    assert canonical_c2_key("body_conv_fcn8_b") == "roi_heads.densepose_head.body_conv_fcn8.bias"
    assert canonical_c2_key("Index_UV_lowres_w") == "roi_heads.densepose_predictor.index_uv_lowres.weight"
    assert canonical_c2_key("some_unknown_blob_w") is None and canonical_c2_key("lr") is None


@pytest.mark.parametrize("name", ["densepose_rcnn_R_50_FPN_s1x_legacy", "densepose_rcnn_R_101_FPN_s1x_legacy"])
def test_caffe2_pickle_roundtrip(tmp_path, name):
    if name.startswith("densepose_rcnn_R_101") and name.endswith("legacy"):
        cfg = get_config("densepose_rcnn_R_50_FPN_s1x_legacy", TINY_OPTS + ["MODEL.RESNETS.DEPTH", 101])
    else:
        cfg = get_config(name, TINY_OPTS)
    state = make_synthetic_state(cfg, 21)
    for k in state:     # Detectron1 checkpoints carry folded affine BN only
        if k.endswith("running_mean"):
            state[k] = np.zeros_like(state[k])
        if k.endswith("running_var"):
            state[k] = np.full_like(state[k], 1.0 - 1e-5)
    path = str(tmp_path / "detectron1.pkl")
    with open(path, "wb") as f:
        pickle.dump({"blobs": _to_caffe2_blobs(cfg, state), "cfg": "ignored"}, f, protocol=2)
    with pytest.raises(ValueError):
        load_checkpoint(path)                # needs the variant
    got = load_checkpoint(path, cfg)
    check_state(cfg, got)                    # strict: nothing missing, shapes right
    assert set(got) == set(param_shapes(cfg))
    for k, v in state.items():
        assert np.array_equal(got[k], v), k


def test_unknown_blob_is_an_error():
    cfg = get_config("densepose_rcnn_R_50_FPN_s1x_legacy", TINY_OPTS)
    with pytest.raises(ValueError):
        convert_caffe2_blobs({"mask_fcn1_w": np.zeros((4, 4, 3, 3), np.float32)}, param_shapes(cfg))

"""Pins the product's config mapping to the reference's own config tree (container-only: needs /root/reference).

oracle/ref_import.py builds the reference FROM the product's ModelConfig when goldens are made, so a wrong default in
config.py would poison golden and product alike. Here the direction is reversed: the reference's
get_cfg() + add_densepose_config + merge_from_file(yaml) (export.py:22-34, + SCORE_THRESH_TEST from export.py:15,23-24)
is the source, and every key the product maps or hard-wires must agree with it for all five BASELINE configs."""
import os

import pytest

from oracle.ref_import import REFERENCE_ROOT, _setup_path, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")

CONFIGS = ["densepose_rcnn_R_50_FPN_s1x_legacy", "densepose_rcnn_R_50_FPN_s1x", "densepose_rcnn_R_101_FPN_s1x",
           "densepose_rcnn_R_50_FPN_DL_s1x", "densepose_rcnn_R_101_FPN_DL_s1x"]


def _reference_cfg(name):
    _setup_path()
    from detectron2.config import get_cfg
    from densepose.config import add_densepose_config
    cfg = get_cfg()
    add_densepose_config(cfg)
    cfg.merge_from_file(os.path.join(REFERENCE_ROOT, "configs", name + ".yaml"))
    cfg.merge_from_list(["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.3])   # export.py:15,23-24 (min_score default)
    return cfg


def _get(cfg, dotted):
    node = cfg
    for part in dotted.split("."):
        if part not in node:
            return None, False
        node = node[part]
    return node, True


@pytest.mark.parametrize("name", CONFIGS)
def test_every_mapped_key_equals_the_reference(name):
    from densepose_torchscript_amd.config import _KEYMAP, get_config
    ref, mine = _reference_cfg(name), get_config(name)
    checked = 0
    for key, (field, conv) in _KEYMAP.items():
        v, present = _get(ref, key)
        assert present, "reference has no key %s" % key
        assert getattr(mine, field) == conv(v), (name, key, getattr(mine, field), v)
        checked += 1
    assert checked == len(_KEYMAP) >= 30
    # the same holds when the product reads the reference's yaml itself (with _BASE_ inheritance)
    from_yaml = type(mine).from_yaml(os.path.join(REFERENCE_ROOT, "configs", name + ".yaml"),
                                     ["MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.3])
    assert from_yaml == mine


@pytest.mark.parametrize("name", CONFIGS)
def test_every_hard_wired_semantic_holds_in_the_reference(name):
    """_FIXED_KEYS = what the kernels implement (aligned=False ROIAlign, 1 class, k4 s2 deconv, x2 bilinear ...): the
    reference's effective value of each such key must be one of the implemented ones for every BASELINE config."""
    from densepose_torchscript_amd.config import _FIXED_KEYS
    ref = _reference_cfg(name)
    for key, allowed in _FIXED_KEYS.items():
        v, present = _get(ref, key)
        assert present, "reference has no key %s" % key
        v = list(v) if isinstance(v, (list, tuple)) else v
        assert any(v == (list(a) if isinstance(a, (list, tuple)) else a) for a in allowed), (name, key, v, allowed)

"""Live cross-check oracle == imported reference. Only runs where /root/reference exists (build container)."""
import numpy as np
import pytest
import torch

from oracle.ref_import import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="/root/reference not present (GPU box)")


@pytest.mark.parametrize("name,wseed", [("densepose_rcnn_R_50_FPN_s1x", 7), ("densepose_rcnn_R_50_FPN_s1x_legacy", 8),
                                        ("densepose_rcnn_R_101_FPN_DL_s1x", 9)])
def test_oracle_equals_reference_live(name, wseed):
    from densepose_torchscript_amd.config import TINY_OPTS, get_config
    from densepose_torchscript_amd.weights import make_synthetic_state
    from oracle.ref_cpu import OracleModel
    from oracle.ref_import import build_reference_predictor
    cfg = get_config(name, TINY_OPTS + ["MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", 7])
    state = make_synthetic_state(cfg, wseed)
    ref = build_reference_predictor(cfg, state)
    ora = OracleModel(cfg, state)
    for hw in [(100, 170), (170, 100)]:
        img = torch.from_numpy(np.random.default_rng(hw[0]).integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8))
        a, b = ref(img), ora(img)
        assert set(a) == set(b)
        for k in a:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
            assert torch.equal(a[k], b[k]), k


def test_caffe2_name_conversion_equals_reference():
    """c2_names.convert_caffe2_blobs == the reference's convert_c2_detectron_names + align_and_update_state_dicts
    (c2_model_loading.py:66-204, :207-299) on a Detectron1-style blob dict of the legacy R50 variant."""
    import logging
    from densepose_torchscript_amd.c2_names import convert_caffe2_blobs
    from densepose_torchscript_amd.config import TINY_OPTS, get_config
    from densepose_torchscript_amd.weights import make_synthetic_state, param_shapes
    from oracle.ref_import import _setup_path
    from test_c2_names import _to_caffe2_blobs
    _setup_path()
    from detectron2.checkpoint.c2_model_loading import align_and_update_state_dicts
    logging.disable(logging.CRITICAL)
    try:
        cfg = get_config("densepose_rcnn_R_50_FPN_s1x_legacy", TINY_OPTS)
        state = make_synthetic_state(cfg, 31)
        blobs = {k: v for k, v in _to_caffe2_blobs(cfg, state).items() if not k.endswith("_momentum")}
        shapes = param_shapes(cfg)
        mine = convert_caffe2_blobs(blobs, shapes)
        model_sd = {k: torch.empty(s) for k, s in shapes.items()}
        theirs = align_and_update_state_dicts(model_sd, {k: torch.from_numpy(np.asarray(v)) for k, v in blobs.items()}, c2_conversion=True)
    finally:
        logging.disable(logging.NOTSET)
    theirs = {k: v for k, v in theirs.items() if k in shapes}
    assert set(theirs) <= set(mine)
    assert {k for k in mine if k not in theirs} == {k for k in mine if k.endswith("running_mean") or k.endswith("running_var")}
    for k, v in theirs.items():
        assert np.array_equal(v.numpy(), mine[k]), k

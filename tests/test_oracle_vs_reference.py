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


@pytest.mark.parametrize("name,wseed", [("densepose_rcnn_R_50_FPN_s1x", 17), ("densepose_rcnn_R_50_FPN_s1x_legacy", 18),
                                        ("densepose_rcnn_R_101_FPN_DL_s1x", 19)])
def test_load_checkpoint_reads_the_reference_state_dict(name, wseed, tmp_path):
    """weights.load_checkpoint on what the reference itself would save - `DefaultPredictor.state_dict()` as a torch `.pth`
    (bare and wrapped in {"model": ...}) and as a detectron2-style `.pkl` ({"model", "__author__"}, detection_checkpoint.py:53-64) -
    gives back the canonical state: the TorchScript fork's ModuleList aliases (stages.N / lateral_convs.N / ...) collapse onto
    the canonical names, the `model.` prefix, pixel_mean / pixel_std and the anchor buffers are dropped, values are untouched."""
    import pickle
    from densepose_torchscript_amd.config import TINY_OPTS, get_config
    from densepose_torchscript_amd.weights import check_state, load_checkpoint, make_synthetic_state, param_shapes
    from oracle.ref_import import build_reference_predictor
    cfg = get_config(name, TINY_OPTS)
    state = make_synthetic_state(cfg, wseed)
    sd = build_reference_predictor(cfg, state).state_dict()
    assert len(sd) > len(state)                         # the aliases are really there
    shapes = param_shapes(cfg)
    pth, pth_wrapped, pkl = str(tmp_path / "m.pth"), str(tmp_path / "w.pth"), str(tmp_path / "m.pkl")
    torch.save(sd, pth)
    torch.save({"model": sd, "iteration": 7}, pth_wrapped)
    with open(pkl, "wb") as f:
        pickle.dump({"model": {k: v.numpy() for k, v in sd.items()}, "__author__": "reference state_dict"}, f, protocol=2)
    for path in (pth, pth_wrapped, pkl):
        got = load_checkpoint(path, cfg)
        assert check_state(cfg, got) == []              # nothing missing, nothing mis-shaped, nothing unknown
        assert set(got) == set(shapes) == set(state)
        for k in shapes:
            assert got[k].dtype == np.float32 and np.array_equal(got[k], state[k]), (path, k)

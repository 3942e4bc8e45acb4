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

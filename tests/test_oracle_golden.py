"""The oracle (oracle/ref_cpu.py) replayed against the golden vectors recorded from the reference."""
import numpy as np
import pytest
import torch

from conftest import golden_case_inputs, load_golden
from oracle.ref_cpu import OracleModel, extract_iuv

CPU_CASES = ["tiny_r50_s1x_a", "tiny_r50_s1x_b", "tiny_r50_legacy", "tiny_r101_s1x", "tiny_r50_dl", "tiny_r101_dl",
             "full_r50_s1x_small", "full_r101_s1x_small", "full_r50_dl_p28", "tiny_r101_dl_p28_video"]


@pytest.mark.parametrize("name", CPU_CASES)
def test_oracle_matches_reference_golden(name):
    meta, z = load_golden(name)
    cfg, state, img = golden_case_inputs(meta)
    model = OracleModel(cfg, state)
    out, inter = model(torch.from_numpy(img), want_all=True)
    s = meta["iuv_stride"]
    for k in ("image_size", "pred_boxes", "scores", "pred_classes"):
        np.testing.assert_array_equal(out[k].numpy(), z["out/" + k], err_msg=k)
    for k in ("pred_densepose_coarse_segm", "pred_densepose_fine_segm", "pred_densepose_u", "pred_densepose_v"):
        # same torch build, same ops, same order -> bit-exact
        np.testing.assert_array_equal(out[k].numpy()[:, :, ::s, ::s], z["out/" + k], err_msg=k)
    # part-index argmax (visualizer.py:10-17) is bit-exact
    for i, (labels, uv) in enumerate(extract_iuv(out)):
        np.testing.assert_array_equal(labels.numpy().astype(np.uint8), z["vis/labels_%d" % i])
        np.testing.assert_array_equal(uv.numpy(), z["vis/uv_%d" % i])
    for nm, cs in meta.get("stage_channel_stride", {}).items():   # cases that store a channel-subsampled stage tensor only
        np.testing.assert_array_equal(inter[nm].numpy()[:, ::cs], z["stage/" + nm], err_msg=nm)
    if "stage/p2" in z.files:
        for k in ("p2", "p3", "p4", "p5", "p6", "box_pooled", "box_logits", "box_deltas", "dp_pooled", "dp_head_out"):
            np.testing.assert_array_equal(inter[k].numpy(), z["stage/" + k], err_msg=k)
        np.testing.assert_array_equal(inter["proposal_boxes"].numpy(), z["stage/proposal_boxes"])
        np.testing.assert_array_equal(inter["objectness_logits"].numpy(), z["stage/objectness_logits"])
        if cfg.dp_decoder_on:
            np.testing.assert_array_equal(inter["decoder_out"].numpy(), z["stage/decoder_out"])


def test_deeplab_p28_goldens_exercise_the_real_head_geometry():
    """POOLER_RESOLUTION 28 (densepose/config.py:177): the d = 6 and d = 12 ASPP branches keep all 9 taps on the 28x28 map,
    d = 56 only the centre (pack.py trims the dead ones); full width means GroupNorm groups of 8 / 16 channels."""
    from densepose_torchscript_amd.config import get_config
    for name, gn_width in (("full_r50_dl_p28", (8, 16)), ("tiny_r101_dl_p28_video", (1, 2))):
        meta, z = load_golden(name)
        cfg = get_config(meta["config"], meta["opts"])
        assert cfg.is_deeplab and cfg.dp_pool == 28
        assert (cfg.fpn_out // 32, cfg.dp_head_dim // 32) == gn_width
        P = cfg.dp_pool
        live = lambda d: sum(1 for r in (-d, 0, d) for s in (-d, 0, d) if abs(r) < P and abs(s) < P)  # noqa: E731
        assert (live(6), live(12), live(56)) == (9, 9, 1)
        assert z["out/pred_densepose_fine_segm"].shape[1:] == (25, 28, 28)      # 112 / stride 4
    meta, z = load_golden("tiny_r101_dl_p28_video")
    assert meta["image_hw"] == [1080, 1920] and z["out/image_size"].tolist() == [1080, 1920]


def test_golden_800x1333_is_the_baseline_geometry():
    meta, z = load_golden("full_r50_s1x_800x1333")
    assert meta["image_hw"] == [800, 1333] and meta["config"] == "densepose_rcnn_R_50_FPN_s1x"
    assert z["out/pred_boxes"].shape == (8, 4)
    assert z["out/pred_densepose_fine_segm"].shape == (8, 25, 14, 14)

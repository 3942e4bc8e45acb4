"""ORACLE tooling (build container only): import the UNMODIFIED reference from /root/reference.

Used by oracle/make_goldens.py and by the container-only cross-check tests
(tests/test_oracle_vs_reference.py, skipped when /root/reference is absent - it never exists
on the GPU box). The four missing third-party packages are replaced by oracle/standins/.
"""
import os
import sys
import tempfile

REFERENCE_ROOT = os.environ.get("DENSEPOSE_REFERENCE_ROOT", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "detectron2"))


def _setup_path():
    for p in (REFERENCE_ROOT, os.path.join(_HERE, "standins")):
        if p not in sys.path:
            sys.path.insert(0, p)


# our ModelConfig field -> reference key, for building the reference's cfg from a ModelConfig
def reference_opts(cfg):
    return [
        "INPUT.MIN_SIZE_TEST", cfg.min_size, "INPUT.MAX_SIZE_TEST", cfg.max_size,
        "MODEL.RESNETS.DEPTH", cfg.depth, "MODEL.RESNETS.STEM_OUT_CHANNELS", cfg.stem_out,
        "MODEL.RESNETS.RES2_OUT_CHANNELS", cfg.res2_out, "MODEL.RESNETS.WIDTH_PER_GROUP", cfg.width_per_group,
        "MODEL.FPN.OUT_CHANNELS", cfg.fpn_out,
        "MODEL.RPN.PRE_NMS_TOPK_TEST", cfg.rpn_pre_topk, "MODEL.RPN.POST_NMS_TOPK_TEST", cfg.rpn_post_topk,
        "MODEL.RPN.NMS_THRESH", cfg.rpn_nms_thresh,
        "MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION", cfg.box_pool, "MODEL.ROI_BOX_HEAD.FC_DIM", cfg.box_fc_dim,
        "MODEL.ROI_HEADS.SCORE_THRESH_TEST", cfg.score_thresh, "MODEL.ROI_HEADS.NMS_THRESH_TEST", cfg.nms_thresh,
        "TEST.DETECTIONS_PER_IMAGE", cfg.dets_per_image,
        "MODEL.ROI_DENSEPOSE_HEAD.NAME", cfg.dp_head,
        "MODEL.ROI_DENSEPOSE_HEAD.DECODER_ON", cfg.dp_decoder_on,
        "MODEL.ROI_DENSEPOSE_HEAD.DECODER_CONV_DIMS", cfg.dp_decoder_dims,
        "MODEL.ROI_DENSEPOSE_HEAD.DECODER_NUM_CLASSES", cfg.dp_decoder_classes,
        "MODEL.ROI_DENSEPOSE_HEAD.POOLER_RESOLUTION", cfg.dp_pool,
        "MODEL.ROI_DENSEPOSE_HEAD.CONV_HEAD_DIM", cfg.dp_head_dim,
        "MODEL.ROI_DENSEPOSE_HEAD.NUM_COARSE_SEGM_CHANNELS", cfg.dp_coarse_ch,
    ]


def build_reference_predictor(cfg, state):
    """cfg: densepose_torchscript_amd.config.ModelConfig; state: canonical name -> ndarray.
    Goes through the reference's own yaml + DetectionCheckpointer .pkl path (export.py:22-34)."""
    _setup_path()
    import torch
    from detectron2.config import get_cfg
    from densepose.config import add_densepose_config
    from detectron2.engine.defaults import DefaultPredictor

    sys.path.insert(0, os.path.dirname(_HERE))
    from densepose_torchscript_amd.weights import save_pkl

    rcfg = get_cfg()
    add_densepose_config(rcfg)
    yaml_path = os.path.join(REFERENCE_ROOT, "configs", cfg.name + ".yaml")
    rcfg.merge_from_file(yaml_path)
    rcfg.merge_from_list(reference_opts(cfg))
    rcfg.MODEL.DEVICE = "cpu"
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "w.pkl")
        save_pkl(state, pkl)
        rcfg.MODEL.WEIGHTS = pkl
        rcfg.freeze()
        torch.manual_seed(12345)  # init of anything NOT covered by the checkpoint would differ -> caught by key check
        pred = DefaultPredictor(rcfg).eval()
    return pred

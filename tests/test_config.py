

def test_unsupported_semantics_raise_instead_of_being_ignored():
    """A yaml / override that asks for semantics the kernels do not implement is an error, never a silent wrong answer
    (the reference's own defaults are ROIAlignV2 and 80 classes: densepose/config.py:176, detectron2/config.py:320)."""
    import pytest
    from densepose_torchscript_amd.config import get_config
    name = "densepose_rcnn_R_50_FPN_s1x"
    for key, bad in (("MODEL.ROI_BOX_HEAD.POOLER_TYPE", "ROIAlignV2"), ("MODEL.ROI_DENSEPOSE_HEAD.POOLER_TYPE", "ROIAlignV2"),
                     ("MODEL.ROI_HEADS.NUM_CLASSES", 80), ("MODEL.ROI_DENSEPOSE_HEAD.DECONV_KERNEL", 2),
                     ("MODEL.ROI_DENSEPOSE_HEAD.UP_SCALE", 4), ("MODEL.ROI_DENSEPOSE_HEAD.CONV_HEAD_KERNEL", 5),
                     ("MODEL.ROI_HEADS.NAME", "StandardROIHeads"), ("MODEL.BACKBONE.NAME", "build_resnet_backbone"),
                     ("MODEL.ROI_DENSEPOSE_HEAD.UV_CONFIDENCE.ENABLED", True), ("MODEL.ROI_DENSEPOSE_HEAD.NAME", "DensePoseV9Head"),
                     ("MODEL.RESNETS.STRIDE_IN_1X1", False), ("MODEL.MASK_ON", True)):
        with pytest.raises(ValueError):
            get_config(name, [key, bad])
    # the implemented values are accepted
    cfg = get_config(name, ["MODEL.ROI_BOX_HEAD.POOLER_TYPE", "ROIAlign", "MODEL.ROI_HEADS.NUM_CLASSES", 1,
                            "MODEL.ROI_DENSEPOSE_HEAD.DECONV_KERNEL", 4])
    assert cfg == get_config(name)
    with pytest.raises(KeyError):
        get_config(name, ["MODEL.NOT_A_KEY", 1])

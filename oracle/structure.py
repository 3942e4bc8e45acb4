"""The network structure the oracle walks, restated from the reference - NOT imported from the product package: a block table or a
decoder layout misread in densepose_torchscript_amd/weights.py must not be shared by the checker (test infrastructure, like the rest of
oracle/; tests/test_abi.py holds the separation in both directions)."""
import math


def resnet_blocks(cfg):
    """Bottleneck blocks of build_resnet_backbone (/root/reference/detectron2/modeling/backbone/resnet.py:641-688): four stages res2 ..
    res5 with num_blocks_per_stage[depth] blocks (:641-647: 50 -> [3, 4, 6, 3], 101 -> [3, 4, 23, 3]); a stage's first block has stride
    2 except res2 (:662 first_stride; RES5_DILATION is 1 in every BASELINE config) and a projection shortcut whenever it changes the
    channel count (BottleneckBlock.__init__ :121-137); in / out / bottleneck channels start at STEM_OUT_CHANNELS / RES2_OUT_CHANNELS /
    NUM_GROUPS * WIDTH_PER_GROUP and the last two double per stage (:684-686).
    -> [(stage name, block index, in channels, bottleneck channels, out channels, stride, has projection shortcut)]"""
    n_per_stage = tuple(cfg.blocks_per_stage)
    assert len(n_per_stage) == 4
    in_ch, out_ch, mid_ch = cfg.stem_out, cfg.res2_out, cfg.width_per_group
    table = []
    for idx, stage_idx in enumerate(range(2, 6)):
        first_stride = 1 if idx == 0 else 2
        for b in range(n_per_stage[idx]):
            stride = first_stride if b == 0 else 1
            block_in = in_ch if b == 0 else out_ch
            table.append(("res%d" % stage_idx, b, block_in, mid_ch, out_ch, stride, block_in != out_ch))
        in_ch = out_ch
        out_ch *= 2
        mid_ch *= 2
    return table


def decoder_layout(cfg):
    """Scale heads of the Panoptic-FPN style Decoder (/root/reference/densepose/modeling/roi_heads/roi_head.py:42-68): for every input
    feature, head_length = max(1, log2(feature stride) - log2(common stride)) 3x3 conv + ReLU layers (:45-47), each followed by a
    bilinear x2 up-sampling when the feature's stride is not the common stride (:62-64). Features p2 .. p5 at strides 4 .. 32, common
    stride 4 (densepose/config.py:193 COMMON_STRIDE). -> [(feature name, number of convolutions)]"""
    common = 4
    out = []
    for name, stride in (("p2", 4), ("p3", 8), ("p4", 16), ("p5", 32)):
        out.append((name, max(1, int(math.log2(stride) - math.log2(common)))))
    return out

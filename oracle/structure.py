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


def rounding_point_fusions(cfg, storage, padded_hw):
    """Which rounding points the ENGINE removes in a 16-bit mode - restated here from its design document so that the
    storage-emulating oracle (oracle/ref_storage.py) is NOT told by the engine what it did: a test derives the answer from the configuration and
    the padded frame size alone and fails if the engine decides otherwise.
      * projection shortcuts (resnet.py:189-190): in bf16 / fp16 the shortcut of a stage's first block is K planes of conv3's matrix
        (res3.0 / res4.0 / res5.0: second source of the pointwise kernel; res2.0: inside the fused bottleneck tail) - no rounded shortcut tensor -
        whenever the two K segments are whole 64-byte planes of 16-bit channels: bottleneck width and block input width (padded to 8) both
        multiples of 32 and their sum a multiple of 64 (every full-width model; the tiny test widths only in their widest stages);
      * the decoder's level sum (roi_head.py:71-79): folded into the epilogues of the scale heads' last convolutions (post_res) when those run on the
        weight-stationary 3x3 kernels that implement it: 256 -> 256 channels (FPN width and DECODER_CONV_DIMS both 256) and at least 128 output
        pixels per image at the geometry of the low heads' last convolution (half of p2's; kernel class 10's size line, never a function of
        the batch). Frames are padded to multiples of 32 (rcnn.py:180), so p2's size is even and every low head ends at exactly half of it.
    -> (list of block prefixes with a fused shortcut, decoder_fold)"""
    if storage not in ("bf16", "fp16"):
        return [], False
    a8 = lambda c: (c + 7) // 8 * 8      # noqa: E731  (channel counts are stored padded to multiples of 8)
    fused = ["backbone.bottom_up.%s.%d." % (stage, b) for stage, b, cin, cmid, _, _, sc in resnet_blocks(cfg)
             if sc and a8(cmid) % 32 == 0 and a8(cin) % 32 == 0 and (a8(cmid) + a8(cin)) % 64 == 0]
    hp, wp = padded_hw
    assert hp % 32 == 0 and wp % 32 == 0, (hp, wp)
    h2, w2 = hp // 4, wp // 4
    fold = bool(cfg.dp_decoder_on and cfg.fpn_out == 256 and cfg.dp_decoder_dims == 256 and (h2 // 2) * (w2 // 2) >= 128)
    return fused, fold

"""The engine's A/B switches in ONE place. Every field's default is what the product runs; a caller that wants something else passes
`options=EngineOptions(...)` to DensePosePredictor / Engine. Nothing under densepose_torchscript_amd/ reads the process environment for
them: the command-line tools (bench.py, tools/*.sh A/B runs: `DP_FUSE_PAIR=0 python bench.py`) build their options with
EngineOptions.from_env() explicitly."""
import dataclasses
import os
import typing


@dataclasses.dataclass
class EngineOptions:
    # independent per-level layers on forked streams, bit mask: 1 FPN output convs, 2 RPN levels, 4 decoder scale heads (engine.py).
    # 0 since the end of round 6: with two pipeline lanes and the decoder's side stream the chip is full, and the RPN levels in line are
    # 2 % (R_50) to 4.5 % (R_101) faster than forked, equal at batch 1 (profiles/r6_ab_fork.txt); rounds 3 - 5 ran 2 (+ 3 % then)
    fork_levels: int = 0
    frames_direct: bool = True              # False = stack the frames of a batch first (round 3)
    fuse_shortcut: bool = True              # block-0 projection shortcut as K planes of conv3 (16-bit modes)
    fuse_sc_tail: bool = True               # ... of res2.0 too (stride 1: inside the fused bottleneck tail)
    fuse_pair: bool = True                  # conv3 and the next block's conv1 of res3's plain blocks in one launch
    group_deconv: bool = True               # the predictor's four sub-pixel convolutions in one grouped launch
    split_k_on: bool = True                 # layers with PackedConv.split_k run split
    decoder_fold: bool = True               # 16-bit modes: the decoder's level sum in the conv epilogues (post_res) instead of a merge pass
    decoder_after_rpn_heads: bool = True    # where the decoder's side stream forks (engine._phase_a)
    fuse_resize: bool = True                # device resize at scale != 1 fused with the preprocess
    identity_resize: bool = True            # frames that already have the test size skip the two resize passes
    # RPN levels with at least this many pixels per image run the hidden 3x3 layer on the one-wave-per-SIMD kernel (class 10) and the two
    # 1x1 heads as a second launch, instead of the LDS-ring kernel with the heads in its epilogue; 0 = never (engine_stages.rpn)
    rpn_split_min_hw: int = 0

    # field -> environment variable of the command-line tools (historical names)
    ENV: typing.ClassVar[dict] = {
        "fork_levels": "DP_FORK", "frames_direct": "DP_FRAMES_DIRECT", "fuse_shortcut": "DP_FUSE_SHORTCUT", "fuse_sc_tail": "DP_FUSE_SC_TAIL",
        "fuse_pair": "DP_FUSE_PAIR", "group_deconv": "DP_GROUP_DECONV", "split_k_on": "DP_SPLIT_K", "decoder_fold": "DP_DECODER_FOLD",
        "decoder_after_rpn_heads": "DP_DEC_LATE", "fuse_resize": "DP_FUSE_RESIZE", "identity_resize": "DP_IDENTITY_RESIZE",
        "rpn_split_min_hw": "DP_RPN_SPLIT"}

    @classmethod
    def from_env(cls, env=None):
        """Options for a command-line tool: every DP_<NAME> variable that is set overrides its field (booleans: "0" = off)."""
        env = os.environ if env is None else env
        kw = {}
        for f in dataclasses.fields(cls):
            v = env.get(cls.ENV[f.name])
            if v is not None and v.strip() != "":
                kw[f.name] = int(v) if f.type is int or f.type == "int" else (v.strip() != "0")
        return cls(**kw)

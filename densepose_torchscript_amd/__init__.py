"""MI355X-native DensePose inference engine (drop-in for dajes/DensePose-TorchScript's predictor callable).

    from densepose_torchscript_amd import DensePosePredictor
    predictor = DensePosePredictor("densepose_rcnn_R_50_FPN_s1x", "model_final.pkl", dtype="bf16")
    outputs = predictor(frame_uint8_hwc_bgr)      # dict with the reference's 8 keys

All arithmetic runs in hand-written HIP kernels for gfx950 behind the C ABI in include/densepose_hip.h
(csrc/*.hip -> libdensepose_hip.so). There is no CPU fallback: without the library or a GPU the product raises.
"""
from .config import ModelConfig, TINY_OPTS, get_config  # noqa: F401
from .weights import load_checkpoint, make_synthetic_state, param_shapes, save_pkl  # noqa: F401


def __getattr__(name):  # lazy: importing the package must not require a GPU (config / weights tooling is host-only)
    if name == "DensePosePredictor":
        from .predictor import DensePosePredictor
        return DensePosePredictor
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)

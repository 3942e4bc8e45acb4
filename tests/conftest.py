import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, z


def golden_case_inputs(meta):
    """Re-create (cfg, state, image uint8 HWC) of a golden case from its seeds."""
    from densepose_torchscript_amd.config import get_config
    from densepose_torchscript_amd.weights import make_synthetic_state, state_checksum
    cfg = get_config(meta["config"], meta["opts"])
    state = make_synthetic_state(cfg, meta["weight_seed"])
    assert state_checksum(state) == meta["weights_sha256"], "synthetic weights differ from the ones the golden was made with"
    h, w = meta["image_hw"]
    img = np.random.default_rng(meta["image_seed"]).integers(0, 256, (h, w, 3), dtype=np.uint8)
    return cfg, state, img


ALL_GOLDENS = ["tiny_r50_s1x_a", "tiny_r50_s1x_b", "tiny_r50_legacy", "tiny_r101_s1x", "tiny_r50_dl", "tiny_r101_dl",
               "full_r50_s1x_small", "full_r101_s1x_small", "full_r50_legacy_small", "full_r50_s1x_800x1333", "full_r50_dl_p28", "full_r101_dl_p28_small", "tiny_r101_dl_p28_video"]


@pytest.fixture(scope="session")
def gpu_available():
    import torch
    return torch.cuda.is_available()


class _Policy:
    """Kernel-policy overrides of the HIP library for one test (dp_set_policy; the library never reads the environment)."""

    def __init__(self):
        from densepose_torchscript_amd import lib
        self.lib = lib
        self.defaults = {}

    def set(self, key, value):
        if key not in self.defaults:
            self.defaults[key] = self.lib.get_policy(key)
        self.lib.set_policy(key, int(value))

    def default(self, key):
        if key in self.defaults:
            self.lib.set_policy(key, self.defaults[key])


@pytest.fixture
def policy():
    pol = _Policy()
    yield pol
    for k, v in pol.defaults.items():
        pol.lib.set_policy(k, v)

"""Load-only Checkpointer: file -> dict -> ndarray->tensor -> non-strict load."""
import logging
from collections import namedtuple

import numpy as np
import torch

_Incompat = namedtuple("_Incompat", ["missing_keys", "unexpected_keys", "incorrect_shapes"])


class Checkpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=True, **checkpointables):
        self.model = model
        self.logger = logging.getLogger(__name__)

    def load(self, path, checkpointables=None):
        if not path:
            return {}
        checkpoint = self._load_file(path)
        self._load_model(checkpoint)
        return checkpoint

    def _load_file(self, f):
        return torch.load(f, map_location="cpu")

    def _convert_ndarray_to_tensor(self, state_dict):
        for k in list(state_dict.keys()):
            v = state_dict[k]
            if isinstance(v, np.ndarray):
                state_dict[k] = torch.from_numpy(v)

    def _load_model(self, checkpoint):
        state = checkpoint.pop("model")
        self._convert_ndarray_to_tensor(state)
        model_state = self.model.state_dict()
        incorrect = []
        for k in list(state.keys()):
            if k in model_state and tuple(model_state[k].shape) != tuple(state[k].shape):
                incorrect.append((k, tuple(state[k].shape), tuple(model_state[k].shape)))
                state.pop(k)
        inc = self.model.load_state_dict(state, strict=False)
        return _Incompat(list(inc.missing_keys), list(inc.unexpected_keys), incorrect)

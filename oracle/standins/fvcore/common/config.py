"""Minimal yacs-like CfgNode: attribute-access dict, _BASE_ yaml inheritance,
merge_from_other_cfg / merge_from_list, clone / freeze / defrost.
Written from the public behaviour of fvcore/yacs; only what the reference's
detectron2/config.py:14-93 and export.py:22-33 call."""
import copy
import os
from ast import literal_eval

import yaml

BASE_KEY = "_BASE_"


class CfgNode(dict):
    IMMUTABLE = "__immutable__"
    NEW_ALLOWED = "__new_allowed__"

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        init_dict = {} if init_dict is None else init_dict
        super().__init__()
        for k, v in init_dict.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = type(self)(v)
            dict.__setitem__(self, k, v)
        self.__dict__[CfgNode.IMMUTABLE] = False
        self.__dict__[CfgNode.NEW_ALLOWED] = new_allowed

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__.get(CfgNode.IMMUTABLE, False):
            raise AttributeError("Attempted to set {} on an immutable CfgNode".format(name))
        if isinstance(value, dict) and not isinstance(value, CfgNode):
            value = type(self)(value)
        self[name] = value

    def is_frozen(self):
        return self.__dict__[CfgNode.IMMUTABLE]

    def _set_immutable(self, flag):
        self.__dict__[CfgNode.IMMUTABLE] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_immutable(flag)

    def freeze(self):
        self._set_immutable(True)

    def defrost(self):
        self._set_immutable(False)

    def is_new_allowed(self):
        return self.__dict__[CfgNode.NEW_ALLOWED]

    def set_new_allowed(self, flag):
        self.__dict__[CfgNode.NEW_ALLOWED] = flag

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        new = type(self)()
        for k, v in self.items():
            dict.__setitem__(new, k, copy.deepcopy(v, memo))
        new.__dict__[CfgNode.IMMUTABLE] = self.__dict__[CfgNode.IMMUTABLE]
        new.__dict__[CfgNode.NEW_ALLOWED] = self.__dict__[CfgNode.NEW_ALLOWED]
        return new

    @classmethod
    def _open_cfg(cls, filename):
        return open(filename, "r")

    @classmethod
    def load_yaml_with_base(cls, filename, allow_unsafe=False):
        with cls._open_cfg(filename) as f:
            cfg = yaml.unsafe_load(f) if allow_unsafe else yaml.safe_load(f)

        def merge_a_into_b(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and k in b and isinstance(b[k], dict):
                    merge_a_into_b(v, b[k])
                else:
                    b[k] = v

        if BASE_KEY in cfg:
            base = cfg.pop(BASE_KEY)
            if base.startswith("~"):
                base = os.path.expanduser(base)
            if not os.path.isabs(base):
                base = os.path.join(os.path.dirname(filename), base)
            base_cfg = cls.load_yaml_with_base(base, allow_unsafe=allow_unsafe)
            merge_a_into_b(cfg, base_cfg)
            return base_cfg
        return cfg

    @staticmethod
    def _decode(v):
        if isinstance(v, str):
            try:
                return literal_eval(v)
            except (ValueError, SyntaxError):
                return v
        return v

    @staticmethod
    def _coerce(replacement, original, full_key):
        ot, rt = type(original), type(replacement)
        if rt == ot or original is None:
            return replacement
        for a, b in [(list, tuple), (tuple, list)]:
            if rt == a and ot == b:
                return b(replacement)
        if ot == float and rt == int:
            return float(replacement)
        raise ValueError("Type mismatch ({} vs. {}) for config key: {}".format(ot, rt, full_key))

    def merge_from_other_cfg(self, other):
        self._merge(other, self, [])

    def _merge(self, a, b, key_list):
        for k, v_ in a.items():
            full_key = ".".join(key_list + [k])
            v = copy.deepcopy(v_)
            v = self._decode(v)
            if k in b:
                if isinstance(v, dict):
                    if not isinstance(b[k], CfgNode):
                        dict.__setitem__(b, k, type(self)(v))
                    else:
                        self._merge(v, b[k], key_list + [k])
                else:
                    dict.__setitem__(b, k, self._coerce(v, b[k], full_key))
            elif b.is_new_allowed():
                if isinstance(v, dict) and not isinstance(v, CfgNode):
                    v = type(self)(v)
                dict.__setitem__(b, k, v)
            else:
                raise KeyError("Non-existent config key: {}".format(full_key))

    def merge_from_file(self, cfg_filename, allow_unsafe=False):
        loaded = self.load_yaml_with_base(cfg_filename, allow_unsafe=allow_unsafe)
        self.merge_from_other_cfg(type(self)(loaded))

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            keys = full_key.split(".")
            d = self
            for sub in keys[:-1]:
                d = d[sub]
            v = self._decode(v)
            dict.__setitem__(d, keys[-1], self._coerce(v, d[keys[-1]], full_key))

    def dump(self, **kwargs):
        def to_dict(n):
            if isinstance(n, CfgNode):
                return {k: to_dict(v) for k, v in n.items()}
            return n

        return yaml.safe_dump(to_dict(self), **kwargs)

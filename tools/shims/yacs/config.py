"""CfgNode: the subset of yacs.config.CfgNode the reference touches (/root/reference/config/stereo_human_config.py:1-62,
train_stage2.py:188-199): attribute-style nested dict, merge_from_file (YAML), defrost / freeze / clone, JSON-serialisable."""
import ast
import copy

import yaml


def _decode(v):
    # yacs runs strings from YAML through literal_eval ("None" -> None, "1e-5" -> 1e-05) and keeps them when that fails
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


class CfgNode(dict):
    _FROZEN = "__frozen__"

    def __init__(self, init=None):
        super().__init__()
        object.__setattr__(self, CfgNode._FROZEN, False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if object.__getattribute__(self, CfgNode._FROZEN):
            raise AttributeError("Attempted to set %s to %r, but CfgNode is immutable" % (k, v))
        self[k] = v

    def _set_frozen(self, flag):
        object.__setattr__(self, CfgNode._FROZEN, flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return object.__getattribute__(self, CfgNode._FROZEN)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, CfgNode._FROZEN, self.is_frozen())
        return out

    def _merge(self, other, path):
        for k, v in other.items():
            where = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: %s" % where)
            if isinstance(v, dict):
                if not isinstance(self[k], CfgNode):
                    raise KeyError("config key %s is not a section" % where)
                self[k]._merge(v, path + [k])
            else:
                dict.__setitem__(self, k, _decode(v))

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            self._merge(yaml.safe_load(f) or {}, [])

    def merge_from_list(self, cfg_list):
        assert len(cfg_list) % 2 == 0
        for k, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node, keys = self, k.split(".")
            for kk in keys[:-1]:
                node = node[kk]
            if keys[-1] not in node:
                raise KeyError("Non-existent config key: %s" % k)
            dict.__setitem__(node, keys[-1], _decode(v))

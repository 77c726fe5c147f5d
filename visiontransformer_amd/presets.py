"""Model presets: the `patch:` blocks of the reference's YAML configs (config/{cifar10,cifar100,ImageNet100}/config.yaml)
as data files, and a config object with the access pattern of utils/MyConfig.py (attribute access into nested
blocks :54-57, item access :59-60, set_ :40-41, set_subkey :43-44, values_ :46-47) -- what `creat_VIT(config)` /
`load_VIT(path, config)` read (mymodel/ViT.py:181-194).  The reference's loader merges a directory of YAML files; a
preset here is one file, and `load_config(path_or_dir)` reads either a file or such a directory (default.yaml first,
then the others in name order, nested blocks merged key by key).

creat_VIT never passes dim_head (mymodel/ViT.py:181-193), so every preset builds heads of 8; build ViT-B/16-like
models with the VIT constructor directly."""
import os

import yaml

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "presets")


class Config:
    def __init__(self, data=None):
        object.__setattr__(self, "_data", data if data is not None else {})

    def __getattr__(self, item):
        try:
            v = self._data[item]
        except KeyError:
            raise AttributeError(item) from None
        return Config(v) if isinstance(v, dict) else v

    def __getitem__(self, item):
        return self._data[item]

    def set_(self, key, value):
        self._data[key] = value

    def set_subkey(self, key, subkey, value):
        self._data[key][subkey] = value

    def values_(self):
        return self._data


def _merge(base, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v
    return base


def _read(path):
    with open(path, "rb") as f:
        return yaml.safe_load(f) or {}


def list_presets():
    return sorted(f[:-5] for f in os.listdir(_DIR) if f.endswith(".yaml"))


def load_config(path):
    """A YAML file, or a directory in the reference's layout (default.yaml, then the other .yaml/.yml files)."""
    if os.path.isdir(path):
        data = {}
        names = sorted(n for n in os.listdir(path) if n.endswith((".yaml", ".yml")))
        for n in (["default.yaml"] if "default.yaml" in names else []) + [n for n in names if n != "default.yaml"]:
            _merge(data, _read(os.path.join(path, n)))
        return Config(data)
    return Config(_read(path))


def load_preset(name, **patch_overrides):
    """`load_preset("cifar10_32x4", num_classes=100)`: the named model block, with overrides of its patch.* entries."""
    path = os.path.join(_DIR, name + ".yaml")
    if not os.path.exists(path):
        raise KeyError(f"no preset {name!r}; available: {', '.join(list_presets())}")
    cfg = load_config(path)
    for k, v in patch_overrides.items():
        cfg.set_subkey("patch", k, v)
    return cfg

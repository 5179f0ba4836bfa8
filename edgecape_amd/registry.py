"""Registries with the reference's names so `configs/test/*.py` build unchanged.

Stands in for mmcv.utils.Registry / build_from_cfg and the mmpose builders the reference uses:
POSENETS / HEADS (EdgeCape.py:17, head.py:61, skeleton.py:9), TRANSFORMER
(EdgeCape/models/utils/builder.py:5,13-15), POSITIONAL_ENCODING (positional_encoding.py:11).
"""


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, module=None, force=False):
        def _reg(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def get(self, key):
        return self._modules.get(key)

    def __contains__(self, key):
        return key in self._modules

    def __repr__(self):
        return f"Registry({self.name}, {sorted(self._modules)})"


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict):
        raise TypeError(f"cfg must be a dict, but got {type(cfg)}")
    if "type" not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    typ = args.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**args)


POSENETS = Registry("posenets")
HEADS = Registry("heads")
TRANSFORMER = Registry("Transformer")
POSITIONAL_ENCODING = Registry("position encoding")


def build_posenet(cfg):
    return build_from_cfg(cfg, POSENETS)


def build_head(cfg):
    return build_from_cfg(cfg, HEADS)


def build_transformer(cfg, default_args=None):
    return build_from_cfg(cfg, TRANSFORMER, default_args)


def build_positional_encoding(cfg, default_args=None):
    return build_from_cfg(cfg, POSITIONAL_ENCODING, default_args)

"""Minimal stand-in for mmcv.Config: the reference's configs are plain Python files without
`_base_` inheritance, so `runpy` loads them unchanged (test.py:87-90)."""
import copy
import runpy


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


class Config:
    def __init__(self, d, filename=None):
        object.__setattr__(self, "_cfg", _wrap(d))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(path):
        ns = runpy.run_path(path)
        d = {k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and not hasattr(v, "__spec__")}
        return Config(copy.deepcopy(d), filename=path)

    def merge_from_dict(self, options):
        """--cfg-options style overrides: {'model.keypoint_head.in_channels': 768} (test.py:47-53,89-90)."""
        for key, val in options.items():
            cur = self._cfg
            parts = key.split(".")
            for p in parts[:-1]:
                cur = cur.setdefault(p, ConfigDict())
            cur[parts[-1]] = _wrap(val)

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def __contains__(self, k):
        return k in self._cfg

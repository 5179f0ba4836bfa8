"""TEST INFRASTRUCTURE ONLY (container-only): make the real reference head importable.

The reference (`/root/reference`, orhir/EdgeCape) depends on mmcv / mmpose / fairseq /
torchvision / cv2, none of which exist in this image.  This module registers *minimal*
stand-in modules carrying only the semantics the hot path uses,
then imports the reference's own `head.py`, `encoder_decoder.py`, `skeleton.py`,
`bias_attn.py`, `positional_encoding.py` and `detectors/EdgeCape.py` *unchanged, from
where they lie*.  Nothing from `/root/reference` is copied into this repository and
this module is never imported by the product path, `bench.py` or the `-m gpu` tests
(`/root/reference` does not exist on the GPU box).

Used by `oracle/make_golden.py` (fixture generation) and by the container-only
`tests/test_reference_configs.py` (skipped when `/root/reference` is absent).
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("EDGECAPE_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "EdgeCape", "models"))


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, module=None, force=False):
        if module is not None:
            self.module_dict[name or module.__name__] = module
            return module

        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def __contains__(self, key):
        return key in self.module_dict


def _build_from_cfg(cfg, registry, default_args=None):
    cfg = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            cfg.setdefault(k, v)
    typ = cfg.pop("type")
    cls = registry.get(typ) if isinstance(typ, str) else typ
    if cls is None:
        raise KeyError(f"{typ} is not in the {registry.name} registry")
    return cls(**cfg)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    """Install stubs + import the reference hot-path modules. Returns a namespace."""
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True  # never write __pycache__ into the reference
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    if "EdgeCape.models.keypoint_heads.head" in sys.modules:
        return _namespace()

    # ---- mmcv -----------------------------------------------------------------
    def xavier_init(module, gain=1, bias=0, distribution="normal"):
        if hasattr(module, "weight") and module.weight is not None:
            if distribution == "uniform":
                nn.init.xavier_uniform_(module.weight, gain=gain)
            else:
                nn.init.xavier_normal_(module.weight, gain=gain)
        if hasattr(module, "bias") and module.bias is not None:
            nn.init.constant_(module.bias, bias)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    POSITIONAL_ENCODING = _Registry("position encoding")
    HEADS = _Registry("heads")
    POSENETS = _Registry("posenets")

    mmcv = _mod("mmcv")
    _mod("mmcv.cnn", Conv2d=nn.Conv2d, Linear=nn.Linear, xavier_init=xavier_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.transformer", POSITIONAL_ENCODING=POSITIONAL_ENCODING,
         build_positional_encoding=lambda cfg, default_args=None: _build_from_cfg(cfg, POSITIONAL_ENCODING, default_args))
    _mod("mmcv.runner", BaseModule=BaseModule)
    _mod("mmcv.utils", Registry=_Registry, build_from_cfg=_build_from_cfg)
    _mod("mmcv.image", imwrite=None)
    _mod("mmcv.visualization")
    _mod("mmcv.visualization.image", imshow=None)
    _mod("cv2")
    _mod("fvcore")
    _mod("fvcore.nn")
    _mod("fvcore.nn.weight_init")

    # ---- mmpose ---------------------------------------------------------------
    builder = _mod("mmpose.models.builder", POSENETS=POSENETS, HEADS=HEADS,
                   build_head=lambda cfg: _build_from_cfg(cfg, HEADS),
                   build_posenet=lambda cfg: _build_from_cfg(cfg, POSENETS))
    _mod("mmpose")
    _mod("mmpose.models", HEADS=HEADS, builder=builder)
    _mod("mmpose.models.detectors")

    class BasePose(nn.Module):
        pass
    _mod("mmpose.models.detectors.base", BasePose=BasePose)
    _mod("mmpose.models.utils")

    def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None, warning=True):
        return F.interpolate(input, size, scale_factor, mode, align_corners)
    _mod("mmpose.models.utils.ops", resize=resize)
    _mod("mmpose.core")
    _mod("mmpose.core.evaluation", keypoint_pck_accuracy=None)

    # ---- fairseq / torchvision (bias_attn.py:13-18) ---------------------------
    def fs_softmax(x, dim, onnx_trace=False):
        return F.softmax(x, dim=dim, dtype=torch.float32)
    _mod("fairseq", utils=_mod("fairseq.utils", softmax=fs_softmax))
    _mod("fairseq.modules")

    class FairseqDropout(nn.Module):
        def __init__(self, p, module_name=None):
            super().__init__()
            self.p = p

        def forward(self, x, inplace=False):
            return F.dropout(x, p=self.p, training=self.training) if self.training and self.p > 0 else x
    _mod("fairseq.modules.fairseq_dropout", FairseqDropout=FairseqDropout)
    _mod("fairseq.modules.quant_noise", quant_noise=lambda m, p, bs: m)

    class TvMLP(nn.Sequential):
        # torchvision.ops.MLP(in, hidden_channels): Linear, ReLU, Dropout per hidden, last Linear + Dropout.
        def __init__(self, in_channels, hidden_channels, dropout=0.0):
            layers = []
            d = in_channels
            for h in hidden_channels[:-1]:
                layers += [nn.Linear(d, h), nn.ReLU(), nn.Dropout(dropout)]
                d = h
            layers += [nn.Linear(d, hidden_channels[-1]), nn.Dropout(dropout)]
            super().__init__(*layers)
    _mod("torchvision", ops=_mod("torchvision.ops", MLP=TvMLP))

    # ---- bare EdgeCape packages so no __init__.py chain runs -------------------
    def bare(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_ROOT, rel)]
        sys.modules[name] = m
        return m
    bare("EdgeCape", "EdgeCape")
    bare("EdgeCape.models", "EdgeCape/models")
    utils_pkg = bare("EdgeCape.models.utils", "EdgeCape/models/utils")
    bare("EdgeCape.models.keypoint_heads", "EdgeCape/models/keypoint_heads")
    bare("EdgeCape.models.detectors", "EdgeCape/models/detectors")
    bare("EdgeCape.models.backbones", "EdgeCape/models/backbones")
    bare("EdgeCape.models.utils.post_processing", "EdgeCape/models/utils/post_processing")
    _mod("EdgeCape.models.backbones.adapter", DPT=None)
    _mod("EdgeCape.models.backbones.dino", DINO=None)

    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    b = importlib.import_module("EdgeCape.models.utils.builder")
    utils_pkg.build_transformer = b.build_transformer
    importlib.import_module("EdgeCape.models.utils.positional_encoding")
    # vendored transform_preds (post_transforms.py:150-194) needs cv2 only at import of other fns
    sys.modules["cv2"].INTER_LINEAR = 1
    pt = importlib.import_module("EdgeCape.models.utils.post_processing.post_transforms")
    sys.modules["mmpose.core.post_processing"] = types.ModuleType("mmpose.core.post_processing")
    sys.modules["mmpose.core.post_processing"].transform_preds = pt.transform_preds
    importlib.import_module("EdgeCape.models.utils.bias_attn")
    importlib.import_module("EdgeCape.models.keypoint_heads.encoder_decoder")
    importlib.import_module("EdgeCape.models.keypoint_heads.skeleton")
    importlib.import_module("EdgeCape.models.keypoint_heads.head")
    return _namespace()


def _namespace():
    ns = types.SimpleNamespace()
    ns.head = sys.modules["EdgeCape.models.keypoint_heads.head"]
    ns.encdec = sys.modules["EdgeCape.models.keypoint_heads.encoder_decoder"]
    ns.skeleton = sys.modules["EdgeCape.models.keypoint_heads.skeleton"]
    ns.bias_attn = sys.modules["EdgeCape.models.utils.bias_attn"]
    ns.posenc = sys.modules["EdgeCape.models.utils.positional_encoding"]
    ns.post = sys.modules["EdgeCape.models.utils.post_processing.post_transforms"]
    ns.HEADS = sys.modules["mmpose.models"].HEADS
    ns.POSENETS = sys.modules["mmpose.models.builder"].POSENETS
    return ns


def install_datasets(metric_fns):
    """Import the reference's host-side data path - EdgeCape/datasets/pipelines/top_down_transform.py (MSRA targets, affine
    warp geometry) and EdgeCape/datasets/datasets/mp100/test_dataset.py + test_base_dataset.py (episode pairing, evaluate /
    _report_metric) - unchanged, from where they lie.  Stand-ins, all documented in the fixtures' meta:
      * cv2.getAffineTransform: the 2x3 solution of the three point pairs in float64 (cv2 is absent; this is the ONE cv2 call
        on the geometry path - cv2.warpAffine, the pixel path, is NOT emulated);
      * json_tricks -> the standard json module (the reference only dumps plain lists / floats);
      * mmpose.core.evaluation.top_down_eval: `metric_fns` (the caller passes edgecape_amd.evaluation's restatement of the
        published mmpose 0.29 functions - mmpose itself is absent, so only the reference's PLUMBING around them is pinned);
      * registries / Compose / DataContainer / COCO: inert placeholders (never called by the pinned functions)."""
    import json
    import numpy as np
    ns = install()

    def get_affine_transform_3pt(src, dst):
        A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
        return np.linalg.solve(A, np.asarray(dst, np.float64)).T
    sys.modules["cv2"].getAffineTransform = get_affine_transform_3pt
    PIPELINES, DATASETS = _Registry("pipeline"), _Registry("dataset")
    _mod("mmcv.fileio")
    sys.modules["mmcv"].fileio = sys.modules["mmcv.fileio"]
    _mod("mmcv.parallel", DataContainer=None)
    _mod("mmpose.datasets", DATASETS=DATASETS)
    _mod("mmpose.datasets.builder", PIPELINES=PIPELINES, DATASETS=DATASETS)
    _mod("mmpose.datasets.pipelines", Compose=lambda p: p)
    pp = sys.modules["mmpose.core.post_processing"]
    for fn in ("affine_transform", "get_affine_transform"):
        setattr(pp, fn, getattr(ns.post, fn))
    for fn in ("fliplr_joints", "get_warp_matrix", "warp_affine_joints"):
        setattr(pp, fn, None)
    _mod("mmpose.core.evaluation.top_down_eval", **metric_fns)
    _mod("json_tricks", dump=json.dump, load=json.load, dumps=json.dumps, loads=json.loads)
    _mod("xtcocotools")
    _mod("xtcocotools.coco", COCO=None)

    def bare(name, rel):
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF_ROOT, rel)]
        sys.modules[name] = m
    bare("EdgeCape.datasets", "EdgeCape/datasets")
    bare("EdgeCape.datasets.pipelines", "EdgeCape/datasets/pipelines")
    bare("EdgeCape.datasets.datasets", "EdgeCape/datasets/datasets")
    bare("EdgeCape.datasets.datasets.mp100", "EdgeCape/datasets/datasets/mp100")
    ns.pipe_post = importlib.import_module("EdgeCape.datasets.pipelines.post_transforms")
    ns.pipe = importlib.import_module("EdgeCape.datasets.pipelines.top_down_transform")
    ns.test_dataset = importlib.import_module("EdgeCape.datasets.datasets.mp100.test_dataset")
    return ns


def import_detector(backbone_factory):
    """Import the reference detector with torch.hub.load patched to `backbone_factory(name)`.

    EdgeCape.py:35-36 calls torch.hub.load('facebookresearch/dinov2', pretrained) which needs
    the network; the oracle's own DINOv2 restatement is substituted.
    """
    import torch
    ns = install()
    if "EdgeCape.models.detectors.EdgeCape" not in sys.modules:
        importlib.import_module("EdgeCape.models.detectors.EdgeCape")
    ns.detector = sys.modules["EdgeCape.models.detectors.EdgeCape"]
    torch.hub.load = lambda repo, name, *a, **k: backbone_factory(name)
    return ns


def load_reference_config(name="configs/test/1shot_split1.py"):
    import runpy
    return runpy.run_path(os.path.join(REF_ROOT, name))

"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from edgecape_amd import _lib, build
    build.build()
    return _lib.load()


def test_exports_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "edgecape_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:const char\*|int)\s+(ec_[a-z0-9_]+)\s*\(", hdr, re.M))
    from edgecape_amd import _lib
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/edgecape_hip.h but not exported"
    assert lib.ec_version() >= 1


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from edgecape_amd import _lib
    cfg = _lib.EcConfig(embed_dim=384, depth=12, num_heads=6, image_size=224, patch=14, num_kpts=100, d_model=256,
                        nhead=8, enc_layers=3, dec_layers=3, skel_layers=3, ffn_dim=384, skel_ffn_dim=384, max_hops=4,
                        heatmap_size=64, max_shots=1, max_batch=2, backbone_precision=0, head_precision=0)
    h = C.c_void_p()
    rc = lib.ec_create(C.byref(cfg), C.byref(h))
    assert rc == -5 and b"no CPU fallback" in lib.ec_last_error()
    from edgecape_amd.engine import HipEngine
    with pytest.raises(_lib.EdgeCapeHipError):
        HipEngine({}, "dinov2_vits14")


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "edgecape_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f"{f} reaches into oracle/"
                assert "/root/reference" not in src, f"{f} reads the reference tree"

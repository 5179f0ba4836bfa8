"""ctypes binding of libedgecape_hip.so (include/edgecape_hip.h).

There is deliberately NO fallback: if the library is missing or no MI355X is visible the import /
the first call raises.  torch is imported first so that the library binds to the HIP runtime already
loaded by PyTorch-ROCm (same SONAME libamdhip64.so.7): device pointers and streams are then shared.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see docstring)

from . import build as _build

_LIB = None

EC_F32, EC_BF16, EC_BF16X3, EC_F16, EC_MIXED, EC_F16X2 = 0, 1, 2, 3, 4, 5
EC_ABI_VERSION = 6   # include/edgecape_hip.h EC_ABI_VERSION: bumped whenever a struct layout, an enum value, the entry points or a signature changes
EC_DT_F32, EC_DT_F16, EC_DT_BF16, EC_DT_F64 = 0, 1, 2, 3
EC_LAYOUT_TOKENS, EC_LAYOUT_NCHW = 0, 1

EXPORTS = ["ec_last_error", "ec_version", "ec_create", "ec_destroy", "ec_load_tensor", "ec_set_pos_embed", "ec_finalize",
           "ec_backbone", "ec_head", "ec_forward", "ec_forward_pipelined", "ec_pipeline_flush", "ec_support_create", "ec_support_destroy", "ec_support_encode", "ec_forward_cached", "ec_forward_episodes", "ec_preprocess_images", "ec_preprocess_images_cv2", "ec_msra_targets", "ec_debug_read", "ec_profile", "ec_profile_read", "ec_op_linear", "ec_op_linear_h16", "ec_op_linear_x2", "ec_op_gemm_bench", "ec_op_bgemm", "ec_op_layernorm",
           "ec_op_attention", "ec_op_chain", "ec_abi_sizes"]


class EcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "embed_dim", "depth", "num_heads", "image_size", "patch", "num_kpts", "d_model", "nhead", "enc_layers",
        "dec_layers", "skel_layers", "ffn_dim", "skel_ffn_dim", "max_hops", "heatmap_size", "max_shots", "max_batch",
        "backbone_precision", "head_precision", "image_width", "gt_skeleton", "no_attn_bias")]


class EcOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "output_kpts_dev", "initial_proposals_dev", "similarity_map_dev", "adj_dev", "attn_adj_dev", "out_points_dev")]


class EdgeCapeHipError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def load():
    """Load (once) and return the ctypes library; raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise EdgeCapeHipError(
            f"{path} not found: build it with `python -m edgecape_amd.build` (needs hipcc). "
            "The EdgeCape hot path has no CPU/PyTorch fallback.")
    if _build.needs_build():
        # a prebuilt library from other sources would be called with mismatched struct layouts: rebuild where hipcc exists
        # (build container, GPU box), refuse otherwise.  build_locked() serialises the ranks of a torchrun job on a lock file
        # and re-checks the stamp once it holds the lock, so exactly one of them compiles.
        try:
            _build.build_locked(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise EdgeCapeHipError(f"{path} is stale (sources changed since it was built) and could not be rebuilt: {e}") from e
    lib = C.CDLL(path)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.ec_last_error.restype = C.c_char_p
    lib.ec_version.restype = ci
    if lib.ec_version() != EC_ABI_VERSION:
        raise EdgeCapeHipError(f"{path}: ABI version {lib.ec_version()} != binding version {EC_ABI_VERSION}")
    lib.ec_abi_sizes.argtypes = [C.POINTER(ci), C.POINTER(ci)]
    lib.ec_abi_sizes.restype = ci
    sc, so = ci(), ci()
    lib.ec_abi_sizes(C.byref(sc), C.byref(so))
    if (sc.value, so.value) != (C.sizeof(EcConfig), C.sizeof(EcOutputs)):
        raise EdgeCapeHipError(f"{path}: struct sizes {sc.value}/{so.value} differ from the ctypes mirrors "
                               f"{C.sizeof(EcConfig)}/{C.sizeof(EcOutputs)}")
    lib.ec_create.argtypes = [C.POINTER(EcConfig), C.POINTER(vp)]
    lib.ec_destroy.argtypes = [vp]
    lib.ec_load_tensor.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), ci, ci]
    lib.ec_set_pos_embed.argtypes = [vp, vp, C.c_int64, C.c_int64]
    lib.ec_finalize.argtypes = [vp]
    lib.ec_backbone.argtypes = [vp, vp, ci, vp, ci, vp]
    lib.ec_head.argtypes = [vp, vp, C.POINTER(vp), ci, C.POINTER(vp), vp, vp, vp, ci, ci, vp, C.POINTER(EcOutputs)]
    lib.ec_forward.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, ci, ci, vp, C.POINTER(EcOutputs)]
    lib.ec_forward_pipelined.argtypes = lib.ec_forward.argtypes
    lib.ec_pipeline_flush.argtypes = [vp, vp]
    lib.ec_support_create.argtypes = [vp, ci, C.POINTER(vp)]
    lib.ec_support_destroy.argtypes = [vp]
    lib.ec_support_encode.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, ci, ci, vp]
    lib.ec_forward_cached.argtypes = [vp, vp, vp, vp, ci, vp, C.POINTER(EcOutputs)]
    lib.ec_forward_episodes.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, vp, vp, ci, ci, vp, vp, ci, vp, C.POINTER(EcOutputs), ci]
    lib.ec_preprocess_images.argtypes = [C.POINTER(vp), vp, vp, vp, ci, ci, vp, vp, vp, vp]
    lib.ec_preprocess_images_cv2.argtypes = [C.POINTER(vp), vp, vp, vp, ci, ci, vp, vp, vp, vp]
    lib.ec_msra_targets.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
    lib.ec_debug_read.argtypes = [vp, C.c_char_p, vp, C.c_int64, C.POINTER(C.c_int64)]
    lib.ec_profile.argtypes = [vp, ci, ci]
    lib.ec_profile_read.argtypes = [vp, C.POINTER(cf), C.POINTER(ci)]
    lib.ec_op_linear.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ec_op_linear_h16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.ec_op_linear_x2.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(cf)]
    lib.ec_op_gemm_bench.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, C.POINTER(cf)]
    lib.ec_op_bgemm.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, vp]
    lib.ec_op_layernorm.argtypes = [vp, vp, vp, vp, ci, ci, cf, vp]
    lib.ec_op_attention.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    lib.ec_op_chain.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, ci, vp, vp, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, vp]
    for n in EXPORTS:
        if n not in ("ec_last_error", "ec_version", "ec_abi_sizes"):
            getattr(lib, n).restype = ci
    _LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise EdgeCapeHipError(f"libedgecape_hip error {rc}: {load().ec_last_error().decode()}")


def current_stream():
    """hipStream_t of torch's current stream as an integer for c_void_p."""
    return torch.cuda.current_stream().cuda_stream

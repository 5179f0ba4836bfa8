"""Input pipeline of the reference's test configs, host geometry + on-device pixels.

configs/test/*.py `test_pipeline`: LoadImageFromFile -> TopDownAffineFewShot -> ToTensor -> NormalizeTensor ->
TopDownGenerateTargetFewShot(sigma=1).  The per-sample geometry (3-point affine from center/scale/rotation,
EdgeCape/models/utils/post_processing/post_transforms.py:197-252; keypoint warp :255-270) is a few flops and stays on the
host; the per-pixel work (warp + normalise, MSRA heatmaps) runs in libedgecape_hip.so (ec_preprocess_images, ec_msra_targets).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def get_affine_transform(center, scale, rot, output_size, shift=(0., 0.), inv=False):
    """The 2x3 matrix of the reference's crop warp (post_transforms.py:197-252), in closed form.

    The reference builds three point pairs - the (shifted) box centre, a point half a box width "above" it rotated by `rot`, and
    a third obtained by a quarter turn of that segment - maps them to the output centre, the point half an OUTPUT width above it
    and its quarter turn, and lets cv2.getAffineTransform solve for the matrix.  Both triangles are right isosceles with the
    same orientation, so the solution is the similarity
        q = s * R(-rot) * (p - c) + d0,     s = output_w / (200 * scale_x),  c = center + 200 * scale * shift,  d0 = output / 2
    (pixel_std = 200).  inv=True returns the inverse map p = R(rot) * (q - d0) / s + c (the reference swaps the triangles)."""
    center, scale = np.asarray(center, np.float64), np.asarray(scale, np.float64)
    assert center.shape == (2,) and scale.shape == (2,) and len(output_size) == 2 and len(shift) == 2
    box = scale * 200.0
    c = center + box * np.asarray(shift, np.float64)
    d0 = np.array([output_size[0] * 0.5, output_size[1] * 0.5])
    s = output_size[0] / box[0]
    th = np.pi * rot / 180.0
    cs, sn = np.cos(th), np.sin(th)
    if inv:
        L = np.array([[cs, -sn], [sn, cs]]) / s          # R(rot) / s
        return np.concatenate([L, (c - L @ d0)[:, None]], 1)
    L = np.array([[cs, sn], [-sn, cs]]) * s              # s * R(-rot)
    return np.concatenate([L, (d0 - L @ c)[:, None]], 1)


def warp_points(pts, trans_mat):
    """Apply a 2x3 affine matrix to points [..., 2] (post_transforms.py:255-270, vectorised)."""
    pts = np.asarray(pts, np.float64)
    M = np.asarray(trans_mat, np.float64)
    return pts @ M[:, :2].T + M[:, 2]


def affine_transform(pt, trans_mat):
    assert len(pt) == 2
    return warp_points(pt, trans_mat)


def gaussian_7x7(sigma=1):
    """The float32 patch of _msra_generate_target (top_down_transform.py:180-186), computed exactly as the reference does."""
    size = 2 * sigma * 3 + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    x0 = y0 = size // 2
    return np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2)).astype(np.float32)


def preprocess_images(images, centers, scales, image_size, rotations=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """images: list of RGB uint8 HWC arrays/tensors (any sizes).  Returns (img [n,3,S,S] fp32 cuda, trans list of 2x3
    src->dst matrices for warping the keypoints with `affine_transform`)."""
    lib = _lib.load()
    n = len(images)
    dev = [torch.as_tensor(np.ascontiguousarray(im) if isinstance(im, np.ndarray) else im).to("cuda", torch.uint8).contiguous()
           for im in images]
    hw = np.array([[d.shape[0], d.shape[1]] for d in dev], np.int32)
    trans, inv = [], np.zeros((n, 6), np.float32)
    for i in range(n):
        r = 0. if rotations is None else rotations[i]
        trans.append(get_affine_transform(centers[i], scales[i], r, (image_size, image_size)))
        inv[i] = get_affine_transform(centers[i], scales[i], r, (image_size, image_size), inv=True).reshape(6)
    out = torch.empty(n, 3, image_size, image_size, device="cuda")
    ptrs = (C.c_void_p * n)(*[d.data_ptr() for d in dev])
    m, s = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    _lib.check(lib.ec_preprocess_images(ptrs, hw.ctypes.data, None, inv.ctypes.data, n, image_size, m.ctypes.data, s.ctypes.data,
                                        out.data_ptr(), _lib.current_stream()))
    torch.cuda.current_stream().synchronize()      # `dev`, `inv` are released on return
    return out, trans


def msra_targets(joints, visible, image_size, heatmap_size=64, sigma=1):
    """joints [n,K,2] (model-input pixels), visible [n,K] -> (target [n,K,hm,hm], target_weight [n,K,1]) on the device."""
    lib = _lib.load()
    j = torch.as_tensor(joints, dtype=torch.float32).cuda().contiguous()
    v = torch.as_tensor(visible, dtype=torch.float32).cuda().contiguous()
    n, K = j.shape[0], j.shape[1]
    target = torch.empty(n, K, heatmap_size, heatmap_size, device="cuda")
    weight = torch.empty(n, K, 1, device="cuda")
    g = np.ascontiguousarray(gaussian_7x7(sigma))
    _lib.check(lib.ec_msra_targets(j.data_ptr(), v.data_ptr(), n, K, image_size, heatmap_size, sigma, g.ctypes.data,
                                   target.data_ptr(), weight.data_ptr(), _lib.current_stream()))
    return target, weight

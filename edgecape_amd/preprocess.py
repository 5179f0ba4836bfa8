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


def _triangle(p0, p1):
    """[p0, p1, p1 rotated a quarter turn about... ] as the reference's float32 3-point set: the third point is p1 + perp(p0 - p1),
    every coordinate rounded to float32 when it is stored (post_transforms.py:49-57, _get_3rd_point :83-102)."""
    t = np.zeros((3, 2), np.float32)
    t[0], t[1] = p0, p1
    d = t[0] - t[1]                                   # float32 arithmetic on the rounded points, as in the reference
    t[2] = t[1] + np.array([-d[1], d[0]], np.float32)
    return t


def _solve_affine(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix with M @ [x, y, 1] = [u, v] for three point pairs, solved in float64 from the
    float32 points (OpenCV converts to double and solves the 6x6 system; the two rows decouple into two 3x3 systems)."""
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, dst.astype(np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=(0., 0.), inv=False):
    """The 2x3 matrix of the reference's crop warp (post_transforms.py:197-252), with the reference's arithmetic.

    Three point pairs - the (shifted) box centre, the point half a box width "above" it rotated by `rot`, and the quarter turn of
    that segment - are mapped to the output centre, the point half an OUTPUT width above it and its quarter turn (pixel_std = 200).
    The points are stored as float32 exactly where the reference stores them, then solved in float64 as cv2.getAffineTransform
    does: the matrix equals the reference's to rounding of the solve (~1e-12), so a joint warped by it lands in the same heatmap
    cell.  (The triangles are right isosceles, i.e. the map is the similarity q = s R(-rot)(p - c) + d0 with s = output_w /
    (200 scale_x); the closed form differs from the reference by the float32 rounding of the points, up to ~5e-3 px.)"""
    center, scale = np.asarray(center), np.asarray(scale)
    assert center.shape == (2,) and scale.shape == (2,) and len(output_size) == 2 and len(shift) == 2
    box = scale * 200.0
    th = np.pi * rot / 180
    sn, cs = np.sin(th), np.cos(th)
    up = -0.5 * box[0]                                            # (0, up) rotated by `rot`
    # the reference's summation order (post_transforms.py:50-51): src[0] = center + box*shift, src[1] = (center + src_dir) + box*shift -
    # with a non-zero shift another order can differ by an ulp in float64 and flip the float32 rounding of the stored point
    sh = box * np.asarray(shift)
    src = _triangle(center + sh, center + np.array([-up * sn, up * cs]) + sh)
    d0 = np.array([output_size[0] * 0.5, output_size[1] * 0.5])
    dst = _triangle(d0, d0 + np.array([0., output_size[0] * -0.5]))
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


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


def preprocess_images(images, centers, scales, image_size, rotations=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, interpolation="cv2"):
    """images: list of RGB uint8 HWC arrays/tensors (any sizes).  Returns (img [n,3,S,S] fp32 cuda, trans list of 2x3
    src->dst matrices for warping the keypoints with `affine_transform`).

    interpolation="cv2" (default): the reference's pixels - cv2.warpAffine's fixed-point INTER_LINEAR on uint8, rounded to uint8,
    then ToTensor / NormalizeTensor (ec_preprocess_images_cv2).  "float": exact float bilinear on un-rounded values."""
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
    if interpolation == "cv2":
        fwd = np.ascontiguousarray(np.stack(trans).reshape(n, 6), np.float64)
        _lib.check(lib.ec_preprocess_images_cv2(ptrs, hw.ctypes.data, None, fwd.ctypes.data, n, image_size, m.ctypes.data, s.ctypes.data,
                                                out.data_ptr(), _lib.current_stream()))
    elif interpolation == "float":
        _lib.check(lib.ec_preprocess_images(ptrs, hw.ctypes.data, None, inv.ctypes.data, n, image_size, m.ctypes.data, s.ctypes.data,
                                            out.data_ptr(), _lib.current_stream()))
    else:
        raise ValueError("interpolation must be 'cv2' or 'float'")
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

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


def rotate_point(pt, angle_rad):
    sn, cs = np.sin(angle_rad), np.cos(angle_rad)
    return np.array([pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs])


def _get_3rd_point(a, b):
    direction = a - b
    return b + np.array([-direction[1], direction[0]], dtype=np.float32)


def _affine_from_3_points(src, dst):
    """cv2.getAffineTransform: the 2x3 matrix M with M @ [x, y, 1] = dst for three point pairs (float64 solve)."""
    A = np.concatenate([np.asarray(src, np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, np.asarray(dst, np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=(0., 0.), inv=False):
    """post_transforms.py:197-252 (pixel_std = 200)."""
    center, scale = np.asarray(center, np.float32), np.asarray(scale, np.float32)
    assert len(center) == 2 and len(scale) == 2 and len(output_size) == 2 and len(shift) == 2
    scale_tmp = scale * 200.0
    shift = np.array(shift)
    src_w, dst_w, dst_h = scale_tmp[0], output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    src_dir = rotate_point([0., src_w * -0.5], rot_rad)
    dst_dir = np.array([0., dst_w * -0.5])
    src = np.zeros((3, 2), dtype=np.float32)
    src[0, :] = center + scale_tmp * shift
    src[1, :] = center + src_dir + scale_tmp * shift
    src[2, :] = _get_3rd_point(src[0, :], src[1, :])
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir
    dst[2, :] = _get_3rd_point(dst[0, :], dst[1, :])
    return _affine_from_3_points(dst, src) if inv else _affine_from_3_points(src, dst)


def affine_transform(pt, trans_mat):
    """post_transforms.py:255-270."""
    assert len(pt) == 2
    return np.array(trans_mat) @ np.array([pt[0], pt[1], 1.])


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

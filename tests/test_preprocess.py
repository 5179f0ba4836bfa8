"""Input pipeline: host geometry on CPU; device kernels (-m gpu) against numpy restatements of the
reference's pipeline steps (cv2 itself is absent from the image: the warp is compared with exact float bilinear)."""
import numpy as np
import pytest
import torch

from edgecape_amd import preprocess as pp
from edgecape_amd import synth


def test_affine_geometry():
    c, s, S = np.array([150., 90.], np.float32), np.array([1.0, 1.0], np.float32) * 1.25, 256
    T = pp.get_affine_transform(c, s, 0., (S, S))
    Ti = pp.get_affine_transform(c, s, 0., (S, S), inv=True)
    # bbox centre -> output centre; bbox (scale*200 px wide) spans the whole output; inverse composes to identity
    assert np.allclose(pp.affine_transform(c, T), [S / 2, S / 2], atol=1e-4)
    assert np.allclose(pp.affine_transform(c - np.array([125., 0.]), T), [0, S / 2], atol=1e-3)
    A = np.vstack([T, [0, 0, 1]]) @ np.vstack([Ti, [0, 0, 1]])
    assert np.allclose(A, np.eye(3), atol=1e-5)
    # rotation by 90 degrees keeps the centre and the isotropic scale
    R = pp.get_affine_transform(c, s, 90., (S, S))
    assert np.allclose(pp.affine_transform(c, R), [S / 2, S / 2], atol=1e-3)
    assert np.isclose(abs(np.linalg.det(R[:, :2])), (S / 250.) ** 2, rtol=1e-4)
    # decode (detector.py / head.py:363-369) is the inverse map for normalised outputs: x*scale*200/W + c - scale*200/2
    pt = np.array([40., 200.])
    back = pt * (s * 200.0) / S + c - s * 200.0 * 0.5
    assert np.allclose(pp.affine_transform(back, T), pt, atol=1e-3)


def _warp_ref(img, Minv, S, mean, std):
    """cv2.warpAffine(INTER_LINEAR, constant-0 border) in exact float arithmetic + ToTensor + NormalizeTensor."""
    ys, xs = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32), indexing="ij")
    sx = Minv[0, 0] * xs + Minv[0, 1] * ys + Minv[0, 2]
    sy = Minv[1, 0] * xs + Minv[1, 1] * ys + Minv[1, 2]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0
    H, W = img.shape[:2]
    out = np.zeros((S, S, 3), np.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            w = (fx if dx else 1 - fx) * (fy if dy else 1 - fy)
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64)
            out += (w * ok)[..., None] * v
    out = (out / 255.0 - np.asarray(mean)) / np.asarray(std)
    return out.transpose(2, 0, 1).astype(np.float32)


@pytest.mark.gpu
def test_preprocess_images_vs_float_bilinear():
    rng = np.random.default_rng(0)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in [(180, 240), (333, 200), (64, 64)]]
    centers = [np.array([120., 90.]), np.array([80., 200.]), np.array([10., 10.])]      # the last box hangs over the border
    scales = [np.array([1.0, 1.0]), np.array([1.7, 1.7]), np.array([0.6, 0.6])]
    rots = [0., 30., -75.]
    S = 224
    out, trans = pp.preprocess_images(imgs, centers, scales, S, rots)
    out = out.cpu().numpy()
    for i in range(3):
        Minv = pp.get_affine_transform(centers[i], scales[i], rots[i], (S, S), inv=True).astype(np.float32)
        ref = _warp_ref(imgs[i], Minv, S, pp.IMAGENET_MEAN, pp.IMAGENET_STD)
        err = np.abs(out[i] - ref).max()
        print("warp", i, "max err", err)
        assert err < 2e-3          # float32 coordinate rounding x up to 255 levels / 0.225; typical 1e-4
    assert out.shape == (3, 3, S, S) and np.isfinite(out).all()


@pytest.mark.gpu
@pytest.mark.parametrize("image_size", [224, 256, 384])
def test_msra_targets_bit_exact(image_size):
    rng = np.random.default_rng(image_size)
    n, K = 5, 100
    joints = rng.uniform(-20, image_size + 20, (n, K, 2)).astype(np.float32)         # includes fully out-of-bounds patches
    joints[0, 0] = [0, 0]
    joints[0, 1] = [image_size - 1, image_size - 1]
    joints[0, 2] = [image_size / 64 * 10.5 - 1e-3, 17.0]                              # rounding boundary of int(x/stride + 0.5)
    vis = (rng.random((n, K)) > 0.3).astype(np.float32)
    t, w = pp.msra_targets(joints, vis, image_size)
    t, w = t.cpu().numpy(), w.cpu().numpy()
    for b in range(n):
        rt, rw = synth.msra_target_ref64(joints[b], vis[b], image_size)
        assert np.array_equal(t[b], rt) and np.array_equal(w[b], rw), b
    assert (t.reshape(n * K, -1) > 0).sum(1).max() <= 49

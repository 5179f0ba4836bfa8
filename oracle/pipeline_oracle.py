"""TEST INFRASTRUCTURE: CPU restatement of the reference's host-side target generator, for the `-m gpu` tests of ec_msra_targets at
sizes / joint sets beyond the committed fixture.  Pinned against the reference itself: tests/test_oracle_golden.py compares it
bit-for-bit with tests/golden/pre_msra.npz (produced by oracle/make_golden.py from the reference's own
TopDownGenerateTargetFewShot._msra_generate_target).  Never imported by the product path or bench.py's timed legs."""
import numpy as np


def msra_target_ref64(joints_xy, visible, image_size, heatmap_size=64, sigma=1):
    """EdgeCape/datasets/pipelines/top_down_transform.py:165-194 (unbiased_encoding=False) with the reference's scalar types:
    `feat_stride = image_size / [W, H]` is a float64 array (:169), so float32 joint / float64 stride + 0.5 is evaluated in float64
    before int()."""
    K = len(joints_xy)
    W = H = heatmap_size
    target = np.zeros((K, H, W), np.float32)
    weight = np.zeros((K, 1), np.float32)
    tmp = sigma * 3
    feat_stride = np.array([image_size, image_size]) / [W, H]
    size = 2 * tmp + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, None]
    x0 = y0 = size // 2
    g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
    for j in range(K):
        weight[j] = visible[j]
        mu_x = int(np.float64(joints_xy[j][0]) / feat_stride[0] + 0.5)
        mu_y = int(np.float64(joints_xy[j][1]) / feat_stride[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if ul[0] >= W or ul[1] >= H or br[0] < 0 or br[1] < 0:
            weight[j] = 0
        if weight[j] > 0.5:
            g_x = max(0, -ul[0]), min(br[0], W) - ul[0]
            g_y = max(0, -ul[1]), min(br[1], H) - ul[1]
            img_x = max(0, ul[0]), min(br[0], W)
            img_y = max(0, ul[1]), min(br[1], H)
            target[j][img_y[0]:img_y[1], img_x[0]:img_x[1]] = g[g_y[0]:g_y[1], g_x[0]:g_x[1]]
    return target, weight


# ---------------------------------------------------------------------------------------------------------------------
# cv2.warpAffine(uint8, INTER_LINEAR, BORDER_CONSTANT 0) -> ToTensor -> NormalizeTensor: the pixel half of the reference's
# input pipeline (EdgeCape/datasets/pipelines/top_down_transform.py:55-58, configs/test/1shot_split1.py:117-125).
#
# cv2 (opencv-python, un-pinned third-party dependency of the reference) is ABSENT from this image, so this is a restatement of
# OpenCV's PUBLISHED algorithm - the classic fixed-point path of modules/imgproc/src/imgwarp.cpp (cv::warpAffine ->
# WarpAffineInvoker -> remap / remapBilinear<FixedPtCast<int, uchar, 15>>, OpenCV 3.x .. 4.10):
#   1. the forward 2x3 matrix is inverted in float64 by the closed form in cv::warpAffine;
#   2. source coordinates in fixed point with AB_BITS = 10: per column adelta[x] = round(M0*x*1024), bdelta[x] = round(M3*x*1024),
#      per row X0 = round((M1*y + M2)*1024) + 16, Y0 = round((M4*y + M5)*1024) + 16 (round = lrint, half to even; 16 =
#      AB_SCALE / INTER_TAB_SIZE / 2), X = (X0 + adelta[x]) >> 5: 5 fractional bits = 1/32 px;
#   3. the four taps are weighted by the int16 table BilinearTab_i[fy*32 + fx] = (32-fy)(32-fx)*32, ... (sum 32768; the one entry
#      that does not fit a short, 32768 at fx = fy = 0, is stored as 32767 with the missing unit moved to the opposite tap - for
#      8-bit pixels the rounded result is the same as with the exact weight, so the exact weights are used here);
#   4. result = (sum + (1 << 14)) >> 15, saturated to uint8; taps outside the image read the border value 0.
# Parity status: "pinned to the published cv2 algorithm (cv2 absent)" - known-answer tests in tests/test_preprocess.py, and the
# HIP kernel (ec_preprocess_images_cv2) is compared with this function bit for bit.
# ---------------------------------------------------------------------------------------------------------------------
def cv2_invert_affine(M):
    """The dst -> src matrix cv::warpAffine derives from the src -> dst matrix it is given (no WARP_INVERSE_MAP), float64."""
    M = np.array(M, np.float64).reshape(2, 3)
    m0, m1, m2, m3, m4, m5 = M.reshape(-1)
    D = m0 * m4 - m1 * m3
    D = 1.0 / D if D != 0 else 0.0
    a11, a22 = m4 * D, m0 * D
    m0, m1, m3, m4 = a11, m1 * (-D), m3 * (-D), a22
    b1 = -m0 * m2 - m1 * m5
    b2 = -m3 * m2 - m4 * m5
    return np.array([[m0, m1, b1], [m3, m4, b2]], np.float64)


def cv2_warp_affine_linear_u8(src, M, dsize):
    """src uint8 [H, W, C]; M the src -> dst 2x3 matrix; dsize = (width, height).  Returns uint8 [height, width, C]."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 3
    Hs, Ws = src.shape[:2]
    Wd, Hd = int(dsize[0]), int(dsize[1])
    Mi = cv2_invert_affine(M)
    xs, ys = np.arange(Wd, dtype=np.float64), np.arange(Hd, dtype=np.float64)
    adelta = np.rint(Mi[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(Mi[1, 0] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((Mi[0, 1] * ys + Mi[0, 2]) * 1024.0).astype(np.int64) + 16
    Y0 = np.rint((Mi[1, 1] * ys + Mi[1, 2]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)           # saturate_cast<short>
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    acc = np.zeros((Hd, Wd, src.shape[2]), np.int64)
    for dy in (0, 1):
        for dx in (0, 1):
            w = (fy if dy else 32 - fy) * (fx if dx else 32 - fx) * 32
            xx, yy = sx + dx, sy + dy
            ok = (xx >= 0) & (xx < Ws) & (yy >= 0) & (yy < Hs)
            v = src[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)].astype(np.int64)
            acc += (w * ok)[..., None] * v
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def to_tensor_normalize(img_u8, mean, std):
    """torchvision F.to_tensor + F.normalize in their float32 arithmetic: (u8 / 255 - mean) / std, HWC -> CHW."""
    x = img_u8.astype(np.float32) / np.float32(255.0)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))

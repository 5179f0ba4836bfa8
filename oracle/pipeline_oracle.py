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

"""moped3d's DEPTHFILL step (DEPTH_FILL_EXACT_CPU, moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp): the oracle's
restatement against cases worked by hand from the source text (the reference holds no fixture for the step and its
header needs OpenCV: parity unpinned)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import orclib  # noqa: E402

K = np.array([525.0, 525.0, 319.5, 239.5], np.float32)


def make_map(h, w, z):
    d = np.zeros((h, w, 4), np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d[..., 2] = z
    d[..., 0] = (u - K[2]) / K[0] * z
    d[..., 1] = (v - K[3]) / K[1] * z
    d[..., 3] = np.sqrt((d[..., :3] ** 2).sum(-1))
    return d


def test_all_valid_is_untouched():
    d = make_map(32, 48, np.full((32, 48), 1.5, np.float32))
    out, dist, s = orclib.depth_fill(d, K, 4)
    assert s == 4 and np.array_equal(out, d) and not dist.any()


def test_one_hole_takes_the_first_strictly_nearer_arrival():
    # scale 2: the downscaled map samples the even rows / columns.  Downscaled pixel (3, 2) = full pixel (6, 4) is a hole.
    h, w, s = 16, 16, 2
    z = (1.0 + 0.01 * np.arange(h * w, dtype=np.float32)).reshape(h, w)
    d = make_map(h, w, z)
    d[4, 6, 2] = -1.0
    out, dist, used = orclib.depth_fill(d, K, s)
    # seeds arrive in raster order: (2,1) first with sqrt(2)*2, then (3,1) with 1*2 -- strictly nearer, it wins; the
    # left / right / lower neighbours tie with it and a tie never replaces (:229)
    want = z[2 * 1, 2 * 3]
    # the hole is full-resolution pixel (6, 4): NNInterp reads column 6 // 2 = 3 and, one row late (:80-82), row (4 - 1) // 2 = 1:
    # that is the downscaled pixel ABOVE the hole -- valid, so its own depth and distance 0
    assert out[4, 6, 2] == z[2, 6] and dist[4, 6] == 0.0
    # the filled downscaled pixel shows one row lower: full pixel (6, 5) is valid, so nothing is written there either
    assert np.array_equal(out[5], d[5])
    # make the row below a hole too: it reads downscaled (3, 2), the filled pixel
    d2 = d.copy()
    d2[5, 6, 2] = -1.0
    out2, dist2, _ = orclib.depth_fill(d2, K, s)
    assert out2[5, 6, 2] == want and dist2[5, 6] == np.float32(2.0)
    # x, y and the norm of a filled pixel follow its new depth (:249-273)
    x = np.float32((np.float32(6) - K[2]) / K[0]) * want
    y = np.float32((np.float32(5) - K[3]) / K[1]) * want
    assert out2[5, 6, 0] == x and out2[5, 6, 1] == y
    assert out2[5, 6, 3] == np.sqrt(np.float32(np.float32(x * x + y * y) + want * want))


def test_scale_one_returns_only_the_distance_map():
    z = np.full((12, 12), 2.0, np.float32)
    z[5, 5] = -1.0
    d = make_map(12, 12, z)
    out, dist, s = orclib.depth_fill(d, K, 1)
    assert s == 1 and np.array_equal(out, d)          # :336-338: the filled map is assigned to a by-value parameter
    assert dist[5, 5] == 1.0 and dist.sum() == 1.0    # the 4-neighbour above arrives second and is strictly nearer


def test_nothing_valid_nothing_filled():
    d = make_map(16, 16, np.full((16, 16), -1.0, np.float32))
    out, dist, _ = orclib.depth_fill(d, K, 2)
    assert (out[..., 2] == -1.0).all() and (dist == np.float32(1e30)).all()


def test_automatic_scale_follows_the_share_of_holes():
    rng = np.random.default_rng(0)
    for share, want in ((0.05, 1), (0.15, 2), (0.3, 4), (0.5, 8), (0.8, 16)):
        z = np.full((64, 64), 1.0, np.float32)
        z.reshape(-1)[rng.permutation(64 * 64)[:int(share * 64 * 64)]] = -1.0
        assert orclib.depth_fill(make_map(64, 64, z), K, -1)[2] == want


def test_bilinear_of_a_constant_fill_is_that_constant():
    z = np.full((32, 32), 2.5, np.float32)
    z[8:24, 8:24] = -1.0
    out, dist, _ = orclib.depth_fill(make_map(32, 32, z), K, 4, bilinear=True)
    assert np.allclose(out[..., 2], 2.5, rtol=1e-6) and (out[..., 2] >= 0).all()
    nn, dist_nn, _ = orclib.depth_fill(make_map(32, 32, z), K, 4, bilinear=False)
    assert (nn[..., 2] == 2.5).all() and dist_nn.max() > 0 and dist.max() > 0


def test_far_holes_take_the_wavefront_not_the_exact_nearest():
    # a big hole: every pixel ends with a finite distance that is at least the true nearest distance
    rng = np.random.default_rng(3)
    z = rng.uniform(0.5, 3.0, size=(60, 80)).astype(np.float32)
    z[10:50, 15:70] = -1.0
    out, dist, _ = orclib.depth_fill(make_map(60, 80, z), K, 1)
    ys, xs = np.nonzero(z >= 0)
    for (y, x) in ((30, 40), (12, 20), (49, 69)):
        true = np.sqrt(((ys - y) ** 2 + (xs - x) ** 2).min())
        assert true - 1e-4 <= dist[y, x] < 1e29

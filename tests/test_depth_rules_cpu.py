"""Oracle restatement of moped3d's DEPTHFILTER_CPU and MATCH_ADAPTIVE_FLANN_CPU ratio rules
(orclib.depthfilter_keep / adaptive_ratio / adaptive_control_points): hand-worked cases.
PARITY UNPINNED against a reference build (moped3d's steps need OpenCV headers); the GPU path is
compared index-exactly with these functions in tests/test_gpu_depth_rules.py."""
import numpy as np

import orclib

K = np.array([800, 800, 320, 240], np.float32)


def _plane(z, h=480, w=640):
    v, u = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = (u - K[2]) / K[0] * z
    img[..., 1] = (v - K[3]) / K[1] * z
    img[..., 2] = z
    img[..., 3] = np.sqrt((img[..., :3] ** 2).sum(-1))
    return img


def test_get_ratio_is_the_piecewise_function_of_the_source():
    cp = (1.0, 2.0, 0.6, 0.8)      # maxRatioDepth, minRatioDepth, ratioLow, ratioHigh
    r = lambda d, mx=4.0: float(orclib._ratio_at(d, cp, mx))
    assert abs(r(0.5) - 0.7) < 1e-6          # linear ratioLow -> ratioHigh below maxRatioDepth
    assert r(1.0) == np.float32(0.8) and r(1.99) == np.float32(0.8)
    assert abs(r(3.0) - 0.4) < 1e-6          # fades to 0 between minRatioDepth and twice that
    assert r(4.0) == 0.0 and r(4.5) == 0.0 and r(3.0, 2.5) == 0.0     # beyond 2*min / beyond MaximumDepth


def test_adjusted_ratio_blends_with_the_default_depth_by_fill_distance():
    img = _plane(0.5)
    table = np.array([[1.0, 2.0, 0.6, 0.8]], np.float32)
    uv = np.array([[100.5, 100.5], [200.2, 50.9]], np.float32)
    fill = np.zeros((480, 640), np.float32)
    fill[50, 200] = 0.1                                       # = CauchyScale -> weight 1/2
    r, reach = orclib.adaptive_ratio(img, fill, uv, np.array([0, 0]), table)
    assert reach.all()
    assert abs(r[0] - 0.7) < 1e-6                             # measured pixel: ratio at its depth
    assert abs(r[1] - (0.5 * 0.7 + 0.5 * 0.8)) < 1e-6         # half way to the ratio at DefaultDepth = 1 m
    img[100, 100, 2] = 4.5
    r, reach = orclib.adaptive_ratio(img, fill, uv, np.array([0, 0]), table)
    assert not reach[0] and reach[1]                          # "Don't even bother searching"


def test_patch_area_and_density_filter():
    img = _plane(0.8)
    inv, pw, ph = orclib.depth_patch_inv_size(img, K, 64)
    assert (pw, ph) == (10, 8)
    side = 64 / 800 * 0.8
    assert abs(1 / inv[0] - side * side) < 1e-6               # first patch row: 64 x 64 pixels at 0.8 m
    assert abs(1 / inv[7 * pw] - side * side) < 1e-6          # last row: y1 = min(512, width) as the source has it
    # 6 features in patch (2,2), 1 in its neighbour (3,2), 1 far away; one feature = 244 per m^2
    uv = np.array([[130 + i, 130 + i] for i in range(6)] + [[200, 150], [600, 400]], np.float32)
    keep = orclib.depthfilter_keep(img, K, 64, 0.05, uv)      # needs > 500 per m^2
    assert keep.tolist() == [True] * 6 + [True, False]        # the neighbour passes through the dilation
    keep = orclib.depthfilter_keep(img, K, 64, 0.02, uv)      # > 200 per m^2: a lone feature passes
    assert keep.all()
    # per-group filtering (ToFilter = 2): the same points split over two models
    keep = orclib.depthfilter_keep(img, K, 64, 0.05, uv, [0, 3, 8])
    assert keep.tolist() == [True, True, True] + [True, True, True, True, False]
    keep = orclib.depthfilter_keep(img, K, 64, 0.05, uv, [0, 2, 8])
    assert keep.tolist() == [False, False] + [True] * 4 + [True, False]   # 2 features: 488 per m^2 is not enough


def test_nan_and_invalid_depths_never_lower_a_patch_minimum():
    img = _plane(1.0)
    img[:64, :64, 2] = np.nan
    img[10, 10, 2] = 0.5
    inv, pw, ph = orclib.depth_patch_inv_size(img, K, 64)
    assert abs(1 / inv[0] - (64 / 800 * 0.5) ** 2) < 1e-6     # std::min skips the NaNs
    img[:64, :64, 2] = np.nan
    inv, _, _ = orclib.depth_patch_inv_size(img, K, 64)
    assert abs(1 / inv[0] - (64 / 800 * 1e10) ** 2) / (64 / 800 * 1e10) ** 2 < 1e-5   # untouched 1e10


def test_control_points_follow_update():
    cp = orclib.adaptive_control_points([-.05, -.05, -.1], [.05, .05, .1], K, 5000)
    # largest face 0.1 x 0.2 m: sqrt(area) * f / depth = 150 px at 0.754 m (1 % tolerance), 50 px at 2.26 m
    assert abs(cp[0] - 0.754) < 0.02 and abs(cp[1] - 2.26) < 0.06
    # 5000 features: the sigmoid is ~0 -> the lower ends of the ratio ranges
    assert abs(cp[2] - 0.6) < 1e-3 and abs(cp[3] - 0.65) < 1e-3
    few = orclib.adaptive_control_points([-.05, -.05, -.1], [.05, .05, .1], K, 200)
    assert abs(few[2] - 0.75) < 1e-3 and abs(few[3] - 0.8) < 1e-3     # few features -> the upper ends


def test_product_control_points_equal_the_oracles():
    """moped_amd.moped3d (what bench.py and Python hosts use) restates the same Update(): identical tables."""
    from moped_amd import moped3d, synth
    db = synth.make_db(4, 300, seed=9)
    tab = moped3d.ratio_table(db.xyz, db.model_of, db.n_models, K)
    for m in range(4):
        sel = db.model_of == m
        want = orclib.adaptive_control_points(db.xyz[sel].min(0), db.xyz[sel].max(0), K, int(sel.sum()))
        assert np.array_equal(tab[m], want)

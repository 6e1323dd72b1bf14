"""moped3d's DEPTHFILL step on the device (mh_depth_fill) against the oracle's restatement of
DEPTH_FILL_EXACT_CPU (moped3d/libmoped/src/depthfill/DEPTH_FILL_EXACT_CPU.hpp): filled depth maps and distance
maps BIT for bit -- the fill is an order-dependent FIFO wavefront, the device replays its queue."""
import numpy as np
import pytest

import orclib
from moped_amd import capi

pytestmark = pytest.mark.gpu
K = np.array([525.0, 525.0, 319.5, 239.5], np.float32)


def make_map(h, w, z):
    d = np.zeros((h, w, 4), np.float32)
    u, v = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d[..., 2] = z
    d[..., 0] = (u - K[2]) / K[0] * z
    d[..., 1] = (v - K[3]) / K[1] * z
    d[..., 3] = np.sqrt((d[..., :3] ** 2).sum(-1))
    return d


def holes(kind, h, w, rng):
    z = (rng.uniform(0.5, 3.5, size=(h, w)) + 0.3 * np.sin(np.arange(w) / 40.0)[None, :]).astype(np.float32)
    if kind == "sparse":
        z[rng.random((h, w)) < 0.05] = -1.0
    elif kind == "dense":
        z[rng.random((h, w)) < 0.7] = -1.0
    elif kind == "blobs":          # sensor-like: shadows beside objects, a dead border, speckle
        for _ in range(25):
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(5, 70)
            yy, xx = np.ogrid[:h, :w]
            z[(yy - cy) ** 2 + ((xx - cx) * rng.uniform(0.3, 1.0)) ** 2 < r * r] = -1.0
        z[:, :12] = -1.0
        z[-9:, :] = -1.0
        z[rng.random((h, w)) < 0.01] = -1.0
    elif kind == "grid":           # every downscaled pixel next to a valid one: the most seeds a map can have
        z[::2, :] = -1.0
    elif kind == "none":
        pass
    elif kind == "all":
        z[:] = -1.0
    elif kind == "one_valid":
        z[:] = -1.0
        z[h // 2 // 16 * 16, w // 3 // 16 * 16] = 2.0
    elif kind == "nan":            # NaN depths are holes (`>= 0` is false)
        z[rng.random((h, w)) < 0.2] = np.nan
    return make_map(h, w, z)


@pytest.mark.parametrize("kind", ["sparse", "dense", "blobs", "grid", "none", "all", "one_valid", "nan"])
@pytest.mark.parametrize("scale,bilinear", [(8, False), (8, True), (16, False)])
def test_depth_fill_bit_exact(ctx, kind, scale, bilinear):
    rng = np.random.default_rng(hash((kind, scale)) % 1000)
    d = holes(kind, 480, 640, rng)
    want, want_dist, _ = orclib.depth_fill(d, K, scale, bilinear)
    got, got_dist, used = ctx.depth_fill(d, K, scale, bilinear)
    assert used == scale
    assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("h,w,scale", [(240, 320, 4), (483, 642, 8), (60, 80, 1), (100, 36, 3), (960, 1280, 16)])
def test_depth_fill_other_shapes(ctx, h, w, scale):
    rng = np.random.default_rng(h + w)
    d = holes("blobs", h, w, rng)
    for bilinear in (False, True):
        want, want_dist, _ = orclib.depth_fill(d, K, scale, bilinear)
        got, got_dist, _ = ctx.depth_fill(d, K, scale, bilinear)
        assert np.array_equal(got_dist.view(np.uint32), want_dist.view(np.uint32))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_depth_fill_automatic_scale_and_limits(ctx):
    rng = np.random.default_rng(1)
    d = holes("dense", 480, 640, rng)            # 70% holes -> factor 16
    want, want_dist, s = orclib.depth_fill(d, K, -1)
    got, got_dist, used = ctx.depth_fill(d, K, -1)
    assert used == s == 16 and np.array_equal(got, want) and np.array_equal(got_dist, want_dist)
    with pytest.raises(capi.MhError):             # 160 x 120 downscaled pixels: more than the LDS-resident fill holds
        ctx.depth_fill(d, K, 4)
    got2, _, _ = ctx.depth_fill(d, K, 16)         # the context keeps working
    assert np.array_equal(got2, want)


def test_depth_fill_feeds_the_frame_on_the_device(ctx):
    """The maps never leave HBM: filled in place, then handed to the frame's depth lookup."""
    import torch
    rng = np.random.default_rng(2)
    d = holes("blobs", 480, 640, rng)
    want, want_dist, _ = orclib.depth_fill(d, K, 8)
    dev = torch.device("cuda:0")
    t_d = torch.from_numpy(d).to(dev)
    t_f = torch.empty((480, 640), dtype=torch.float32, device=dev)
    assert ctx.depth_fill_dev(t_d.data_ptr(), 640, 480, K, t_f.data_ptr(), 8) == 8
    ctx.depth_fill_status()
    assert np.array_equal(t_d.cpu().numpy(), want) and np.array_equal(t_f.cpu().numpy(), want_dist)
    ctx.frame_set_depth_image(t_d.data_ptr(), t_f.data_ptr(), 640, 480, capi.DEPTH_BACKPROJECTION)
    ctx.frame_set_depth_image(0, 0, 0, 0, 0)

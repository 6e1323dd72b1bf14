"""N2: the SIFT oracle (oracle/sift_oracle.cpp) against the reference's own
libsiftfast build: bit-identical keypoints, order, scale, orientation and descriptors on the
bundled frames (tests/golden/sift_ref_frames.npz, made by oracle/make_golden.py siftref)."""
import os

import numpy as np
import pytest

import orclib

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sift_ref_frames.npz"))


@pytest.mark.parametrize("f", [int(x) for x in GOLD["frames"]])
def test_sift_oracle_is_bit_identical_to_the_reference_build(f):
    xy, so, d = orclib.sift(GOLD[f"gray{f}"])
    assert len(xy) == len(GOLD[f"xy{f}"]) > 500
    assert np.array_equal(xy, GOLD[f"xy{f}"])
    assert np.array_equal(so, GOLD[f"scale_ori{f}"])
    assert np.array_equal(d, GOLD[f"desc{f}"])


def test_sift_oracle_small_and_flat_images():
    flat = np.full((64, 80), 128, np.uint8)
    xy, so, d = orclib.sift(flat)
    assert len(xy) == 0                                   # no extrema on a constant image
    tiny = (np.arange(14 * 14).reshape(14, 14) % 251).astype(np.uint8)
    xy, _, _ = orclib.sift(tiny, double_size=False)       # 14 > 12: one octave is processed
    assert xy.shape[1] == 2
    rng = np.random.default_rng(0)
    g = (rng.random((120, 160)) * 255).astype(np.uint8)
    a = orclib.sift(g)
    b = orclib.sift(g)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and len(a[0]) > 50   # deterministic
    assert np.allclose(np.linalg.norm(a[2], axis=1), 1.0, atol=1e-4)            # unit descriptors

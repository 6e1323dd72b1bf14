"""Oracle restatement of moped3d's CLUSTER_LINKAGE_CPU (oracle/linkage_oracle.cpp): hand-worked
cases.  PARITY UNPINNED against a reference build (moped3d's steps need OpenCV headers); the GPU
kernel is compared with this oracle in tests/test_gpu_linkage.py."""
import numpy as np

import orclib

K = np.array([800, 800, 320, 240], np.float32)


def _plane(z=1.0, h=480, w=640):
    v, u = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = (u - K[2]) / K[0] * z
    img[..., 1] = (v - K[3]) / K[1] * z
    img[..., 2] = z
    img[..., 3] = np.sqrt((img[..., :3] ** 2).sum(-1))
    return img


def _world(img, uv):
    return img[uv[:, 1].astype(int), uv[:, 0].astype(int), :3].copy()


def test_similarity_matrix_on_a_flat_scene():
    """Flat depth: the discontinuity kernel is exp(0) = 1 off the diagonal and exp(1 / (-2 (pi/128)^2)) = 0 on
    it (a single Bresenham sample leaves maxAngleDiff at -1); model == world points make the
    distance-consistency kernel 1; measured pixels (fill distance 0) give w2D = w3D = 0.5."""
    img = _plane()
    uv = np.array([[100, 100], [110, 100], [100, 112], [400, 300]], np.float32)
    world = _world(img, uv)
    cl, Km = orclib.cluster_linkage(uv, world, world, img, np.zeros((480, 640), np.float32), min_pts=0, want_k=True)
    assert np.array_equal(Km, Km.T)
    # sigma2D = mean nearest-neighbour distance = (10 + 10 + 12 + |(400,300)-(110,100)|) / 4
    d = np.sqrt(((uv[:, None] - uv[None]) ** 2).sum(-1)); np.fill_diagonal(d, np.inf)
    s2 = np.float32(d.min(1).astype(np.float32).sum() / 4)
    dw = np.sqrt(((world[:, None] - world[None]) ** 2).sum(-1)); np.fill_diagonal(dw, np.inf)
    s3 = np.float32(dw.min(1).astype(np.float32).sum() / 4)
    k2 = np.exp(-((uv[0] - uv[1]) ** 2).sum() / (2 * s2 * s2))
    k3 = np.exp(-((world[0] - world[1]) ** 2).sum() / (2 * s3 * s3))
    # K3D + BK = k3 + 1, normalised by the matrix maximum (the largest off-diagonal k3 + 1), times K3F = 1, normalised again
    offd = [np.exp(-((world[i] - world[j]) ** 2).sum() / (2 * s3 * s3)) + 1 for i in range(4) for j in range(4) if i != j]
    want = 0.5 * k2 + 0.5 * (k3 + 1) / max(offd)
    assert abs(Km[0, 1] - want) < 1e-5
    assert abs(Km[0, 0] - (0.5 * 1 + 0.5 * 1 / max(offd))) < 1e-5     # diagonal: K2D = 1, K3D = 1 + 0
    # on a flat scene the discontinuity kernel alone keeps every similarity at 0.25 or more: with the
    # shipped cutoff (0.1) everything ends up in one cluster, the far point included
    assert Km.min() > 0.25 and [sorted(c.tolist()) for c in cl] == [[0, 1, 2, 3]]
    cl = orclib.cluster_linkage(uv, world, world, img, np.zeros((480, 640), np.float32), cutoff=0.5, min_pts=0)
    assert [sorted(c.tolist()) for c in cl] == [[0, 1, 2], [3]]


def test_member_order_and_min_pts():
    """Merging appends the absorbed cluster back to front; clusters need MORE than MinPts members."""
    img = _plane()
    img[:, 320:, :3] *= 3.0                                             # a 2 m depth step between the two groups
    rng = np.random.default_rng(3)
    a = rng.uniform([100, 100], [140, 140], (9, 2)).astype(np.float32)
    b = rng.uniform([400, 300], [440, 340], (8, 2)).astype(np.float32)
    uv = np.concatenate([a, b])
    world = _world(img, uv)
    cl = orclib.cluster_linkage(uv, world, world, img, None)            # MinPts 7 -> both groups (9 and 8 > 7)
    assert [sorted(c.tolist()) for c in cl] == [list(range(9)), list(range(9, 17))]
    assert cl[0][0] == 0 and cl[1][0] == 9                              # a cluster starts with its lowest index
    assert not np.array_equal(cl[0], np.sort(cl[0]))                    # ... and is not in index order
    cl8 = orclib.cluster_linkage(uv, world, world, img, None, min_pts=8)
    assert [sorted(c.tolist()) for c in cl8] == [list(range(9))]        # 8 is not MORE than 8
    # a high cutoff leaves every point alone, and singletons are not clusters
    assert orclib.cluster_linkage(uv, world, world, img, None, cutoff=2.0) == []
    assert [len(c) for c in orclib.cluster_linkage(uv, world, world, img, None, cutoff=2.0, min_pts=0)] == [1] * 17


def test_depth_discontinuity_separates_what_the_image_joins():
    """Two groups next to each other in the image but 0.5 m apart in depth, with a depth step
    between them: the discontinuity and 3-D kernels keep them apart."""
    img = _plane(1.0)
    img[:, 320:, :3] *= 1.5
    rng = np.random.default_rng(5)
    a = rng.uniform([280, 200], [315, 240], (10, 2)).astype(np.float32)
    b = rng.uniform([325, 200], [360, 240], (10, 2)).astype(np.float32)
    uv = np.concatenate([a, b])
    world = _world(img, uv)
    cl = orclib.cluster_linkage(uv, world, world, img, None)
    assert [sorted(c.tolist()) for c in cl] == [list(range(10)), list(range(10, 20))]
    flat = _plane(1.0)
    cl = orclib.cluster_linkage(uv, _world(flat, uv), _world(flat, uv), flat, None)
    assert [sorted(c.tolist()) for c in cl] == [list(range(20))]        # same image points on a flat scene: one cluster


def test_degenerate_inputs():
    img = _plane()
    assert orclib.cluster_linkage(np.zeros((0, 2), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), img, None) == []
    one = np.array([[10, 10]], np.float32)
    assert orclib.cluster_linkage(one, _world(img, one), _world(img, one), img, None, min_pts=0)[0].tolist() == [0]
    same = np.array([[10.2, 10.7]] * 9, np.float32)        # identical pixels: sigma = 0, similarities NaN/0 -> no merges
    assert orclib.cluster_linkage(same, _world(img, same), _world(img, same), img, None) == []


def _agglomerate(Km, cutoff, min_pts, ltype):
    """CLUSTER_LINKAGE_CPU's merge loop (:437-540) written down a second time, in Python, over a similarity matrix --
    including what its list handling does: the absorbed cluster's index stays in the list of live indices until the NEXT
    scan reaches it (:456-459: erased there, and the element behind it is skipped as a first index of that scan), so
    earlier first indices still pair with it, through the matrix row the last update gave it.  Linkage of two clusters
    = minimum (0) / maximum (2) of K over their pairs, recomputed from K (minimumLinkage :404-413: 1e20 over an empty
    cluster, maximumLinkage :390-399: -1)."""
    n = len(Km)
    cl = [[i] for i in range(n)]
    live = list(range(n))
    D = Km.astype(np.float32).copy()
    def link(A, B):
        v = [Km[b, a] for a in A for b in B]
        return np.float32((min(v) if ltype == 0 else max(v)) if v else (1e20 if ltype == 0 else -1.0))
    remove = -1
    while True:
        best, pair = np.float32(-1), (0, 0)
        x = 0
        while x < len(live):
            i = live[x]
            if i == remove:
                del live[x]          # erase; the loop's increment then skips the element that moved into this place
                x += 1
                continue
            for j in live[x + 1:]:
                if D[i, j] > best:
                    best, pair = D[i, j], (i, j)
            x += 1
        if best < cutoff:
            break
        i, j = pair
        cl[i] += cl[j][::-1]
        cl[j] = []
        remove = j
        for k in range(n):
            D[i, k] = D[k, i] = link(cl[k], cl[i])
    return [c for c in cl if len(c) > min_pts]


def test_minimum_and_maximum_linkage_against_an_independent_agglomeration():
    """LinkageType 0 / 2 (CLUSTER_LINKAGE_CPU.hpp:506-507, :525; the shipped configuration uses 1): the oracle's clusters,
    members in its order, equal a from-scratch agglomeration over the oracle's own similarity matrix."""
    img = _plane()
    img[:, 320:, :3] *= 2.0
    rng = np.random.default_rng(11)
    uv = np.concatenate([rng.uniform([60, 60], [200, 200], (14, 2)), rng.uniform([380, 250], [520, 400], (12, 2)),
                         rng.uniform([0, 0], [639, 479], (6, 2))]).astype(np.float32)
    world = _world(img, uv)
    fill = np.zeros((480, 640), np.float32)
    seen = set()
    for ltype in (0, 2):
        for cutoff in (0.05, 0.2, 0.45, 0.7):
            got, Km = orclib.cluster_linkage(uv, world, world, img, fill, cutoff=cutoff, min_pts=1, linkage_type=ltype, want_k=True)
            want = _agglomerate(Km, np.float32(cutoff), 1, ltype)
            assert [c.tolist() for c in got] == want, (ltype, cutoff)
            seen.add((ltype, len(want)))
    assert len({n for _, n in seen}) >= 2            # the cutoffs do produce different partitions
    # minimum linkage merges no further than maximum linkage at the same cutoff
    n0 = len(orclib.cluster_linkage(uv, world, world, img, fill, cutoff=0.45, min_pts=0, linkage_type=0))
    n2 = len(orclib.cluster_linkage(uv, world, world, img, fill, cutoff=0.45, min_pts=0, linkage_type=2))
    assert n0 >= n2

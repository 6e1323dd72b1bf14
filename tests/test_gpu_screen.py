"""Two-stage MATCH (csrc/match_screen.hip: f16 matrix-pipe screen + canonical f32 arithmetic on the
candidates) against the exact f32 kernels and the oracle: (idx1, d1, d2) must be the same BITS for
every query, whatever the screen sees -- the screen may only cost time, never change a result.

Adversarial inputs aim at the screen's error bound: rows closer together than f16 can tell apart,
exact duplicates (ties -> lower row; candidate lists overflow -> brute force inside pass C), zero
rows, unnormalised rows and queries, queries f16 cannot hold, DB shards with index_base != 0, query
counts that are not multiples of the 256-query blocks.  (Device-side query counts: the image
frames of tests/test_gpu_image_frame.py go through the two-stage path as well.)"""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu


def _search(c, torch, qn, mode, q_count=None):
    dev = torch.device("cuda:0")
    Q = qn.shape[0]
    tq = torch.from_numpy(np.ascontiguousarray(qn)).to(dev)
    qnorm = torch.from_numpy(orclib.row_norms(qn)).to(dev)
    out = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
    c.match_set_mode(mode)
    c.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, *[o.data_ptr() for o in out])
    c.synchronize()
    c.match_set_mode(-1)
    return [o.cpu().numpy() for o in out]


def _same_bits(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
            and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)))


@pytest.fixture(scope="module")
def env():
    import torch
    c = capi.Context(0)
    yield c, torch
    c.close()


def test_config1_frames_two_stage_equals_exact_kernels_and_oracle(env):
    c, torch = env
    db = synth.make_db(20, 5000)
    dbn = orclib.normalize(db.desc)
    c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    assert c.match_stats(3000)["two_stage"]
    c.match_stats(reset=True)
    for seed in (0, 1):
        fr = synth.make_frame(db, n_vis=3, seed=seed)
        qn = orclib.normalize(fr.desc)
        two = _search(c, torch, qn, 1)
        one = _search(c, torch, qn, 0)
        assert _same_bits(two, one)
        pick = np.sort(np.random.default_rng(seed).choice(len(qn), 64, replace=False))
        oi, o1, o2 = orclib.match_2nn(dbn, qn[pick])
        assert _same_bits([two[0][pick], two[1][pick], two[2][pick]], [oi, o1, o2])
    st = c.match_stats()
    assert st["queries"] == 6000 and st["brute_force_queries"] == 0
    # the screen is selective: tens of rows per query out of 100,000
    assert 2 <= st["candidates"] / st["queries"] < 200


@pytest.mark.parametrize("Q", [1, 255, 256, 257, 1000, 3071])
def test_ragged_query_counts(env, Q):
    c, torch = env
    db = synth.make_db(4, 2000, seed=3)
    dbn = orclib.normalize(db.desc)
    c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    base, _, _ = synth.load_sift_fixture()
    qn = orclib.normalize(base[np.random.default_rng(Q).integers(0, len(base), Q)])
    two = _search(c, torch, qn, 1)
    oi, o1, o2 = orclib.match_2nn(dbn, qn)
    assert _same_bits(two, [oi, o1, o2])


def test_shard_with_index_base_and_ragged_rows(env):
    c, torch = env
    db = synth.make_db(7, 1111, seed=9)            # 7777 rows: the last tile is padded
    dbn = orclib.normalize(db.desc)
    lo = 2222
    c.db_upload(dbn[lo:], db.model_of[lo:], db.xyz[lo:], db.n_models, index_base=lo)
    base, _, _ = synth.load_sift_fixture()
    qn = orclib.normalize(base[:700])
    two = _search(c, torch, qn, 1)
    oi, o1, o2 = orclib.match_2nn(dbn[lo:], qn)
    assert _same_bits(two, [np.where(oi >= 0, oi + lo, -1).astype(np.int32), o1, o2])


def _adversarial_db(rng, n=9000):
    """Rows the f16 screen cannot tell apart: 40 bases, each with 150 copies perturbed by 1e-5 .. 3e-4 (below
    f16's resolution of 5e-4 at these magnitudes), 8 exact duplicates of every base scattered over the rows,
    a block of all-zero rows, a block of rows with norms 0.25 .. 3, the rest unrelated unit rows."""
    base, _, _ = synth.load_sift_fixture()
    b = orclib.normalize(base[rng.choice(len(base), 40, replace=False)])
    rows = []
    for k in range(40):
        amp = 10.0 ** rng.uniform(-5, -3.5, size=(150, 1))
        rows.append(b[k] + amp * rng.normal(size=(150, 128)))
        rows.append(np.repeat(b[k][None], 8, 0))
    near = np.maximum(np.concatenate(rows), 0).astype(np.float32)
    zeros = np.zeros((64, 128), np.float32)
    scaled = orclib.normalize(base[rng.choice(len(base), 400)]) * rng.uniform(0.25, 3.0, size=(400, 1)).astype(np.float32)
    rest = orclib.normalize(base[rng.choice(len(base), n - len(near) - 64 - 400)] +
                            rng.normal(0, 0.05, size=(n - len(near) - 64 - 400, 128)).astype(np.float32).clip(0))
    db = np.concatenate([near, zeros, scaled.astype(np.float32), rest]).astype(np.float32)
    db = db[rng.permutation(len(db))]
    return np.ascontiguousarray(db), b


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_adversarial_rows_inside_the_screens_error_bound(env, seed):
    c, torch = env
    rng = np.random.default_rng(100 + seed)
    db, bases = _adversarial_db(rng)
    n = len(db)
    c.db_upload(db, np.zeros(n, np.int32), np.zeros((n, 3), np.float32), 1)      # rows as they are: not re-normalised
    # queries: the bases themselves (ties between their 8 duplicates: lowest row must win, d1 == d2 == 0-ish),
    # bases + noise at the scale of the perturbations, scaled queries, and a zero query
    qs = [bases, bases + rng.normal(0, 1e-4, bases.shape), bases * 0.5, bases * 2.0, np.zeros((3, 128))]
    qn = np.ascontiguousarray(np.concatenate(qs), np.float32)
    c.match_stats(reset=True)
    two = _search(c, torch, qn, 1)
    one = _search(c, torch, qn, 0)
    oi, o1, o2 = orclib.match_2nn(db, qn)
    assert _same_bits(one, [oi, o1, o2])
    assert _same_bits(two, [oi, o1, o2])
    # ties really occur (a base against its duplicates AND against copies perturbed below f32's resolution of the
    # distance: all exactly 0 in the canonical arithmetic); the oracle's rule -- lowest row -- is what both paths give
    tie = np.nonzero(o1[:40] == o2[:40])[0]
    assert len(tie) >= 30
    st = c.match_stats()
    assert st["queries"] == len(qn)


def test_candidate_overflow_falls_back_to_brute_force(env):
    """6000 rows, 5000 of them the same vector: every query's candidate list overflows."""
    c, torch = env
    base, _, _ = synth.load_sift_fixture()
    rng = np.random.default_rng(7)
    v = orclib.normalize(base[:1])
    db = np.concatenate([np.repeat(v, 5000, 0), orclib.normalize(base[rng.choice(len(base), 1000)])]).astype(np.float32)
    db = np.ascontiguousarray(db[rng.permutation(len(db))])
    c.db_upload(db, np.zeros(len(db), np.int32), np.zeros((len(db), 3), np.float32), 1)
    qn = np.ascontiguousarray(np.concatenate([v, orclib.normalize(base[5:300])]), np.float32)
    c.match_stats(reset=True)
    two = _search(c, torch, qn, 1)
    oi, o1, o2 = orclib.match_2nn(db, qn)
    assert _same_bits(two, [oi, o1, o2])
    assert c.match_stats()["brute_force_queries"] >= 1
    assert two[0][0] == int(np.nonzero((db == v[0]).all(1))[0].min()) and two[1][0] == two[2][0]


def test_queries_f16_cannot_hold(env):
    c, torch = env
    db = synth.make_db(3, 2000, seed=5)
    dbn = orclib.normalize(db.desc)
    c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    base, _, _ = synth.load_sift_fixture()
    qn = orclib.normalize(base[:300]).copy()
    qn[3] *= 1e6                       # beyond f16's range
    qn[7, 5] = np.inf
    qn[11, 0] = np.nan
    qn[13] = 0.0
    qn[17] *= 1e-6                     # deep in f16's subnormal range
    c.match_stats(reset=True)
    with np.errstate(all="ignore"):
        two = _search(c, torch, qn, 1)
        one = _search(c, torch, qn, 0)
    assert _same_bits(two, one)
    # the huge, the inf and the NaN query (the screen cannot hold them) + the 1e-6 one, for which every row is inside the
    # margin of every other (candidate list overflow): brute force inside pass C; the all-zero query -- the same tie --
    # takes the answer the upload prepared for it (its distance to a row is the row's norm term)
    assert c.match_stats()["brute_force_queries"] == 4


def test_db_the_screen_must_refuse(env):
    """A NaN / inf / out-of-range row switches the screen off for that DB (exact kernels run)."""
    c, torch = env
    db = synth.make_db(3, 2000, seed=6)
    dbn = orclib.normalize(db.desc)
    base, _, _ = synth.load_sift_fixture()
    qn = orclib.normalize(base[:300])
    for poison in (np.nan, np.inf, 1e6):
        bad = dbn.copy()
        bad[1234, 17] = poison
        c.db_upload(bad, db.model_of, db.xyz, db.n_models)
        assert not c.match_stats(3000)["two_stage"]
        with np.errstate(all="ignore"):
            two = _search(c, torch, qn, 1)
            one = _search(c, torch, qn, 0)
        assert _same_bits(two, one)
    c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
    assert c.match_stats(3000)["two_stage"]


def test_randomised_shapes_and_value_distributions():
    """scripts/screen_stress.py (36 of its cases here; 1200 were run when the launch policy last changed): DB sizes
    and query counts around the policy's thresholds, unnormalised rows, near-duplicate clusters, integer-valued
    descriptors, shards -- two-stage and exact kernels agree bit for bit."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "screen_stress.py"), "36", "7"],
                         capture_output=True, text=True)
    assert out.returncode == 0 and "36 cases, 0 mismatches" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]

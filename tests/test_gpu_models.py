"""N3 on the GPU: a database uploaded from model files / a mapped `.mopeddb` container
(normalised on the device) matches bit for bit like one normalised by the oracle and
uploaded from arrays; model blocks give the shards of SURVEY 8(e)."""
import os

import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("models")
    db = synth.make_db(4, 600, seed=77)
    paths, xyz, desc = [], [], []
    for m in range(4):
        rows = db.model_of == m
        p = str(d / f"m{m}.moped.xml")
        rx, rd = synth.write_model_xml(p, f"model{m}", db.xyz[rows], db.desc[rows], full_export=(m == 1), seed=m)
        paths.append(p)
        xyz.append(rx)
        desc.append(rd)
    return paths, np.concatenate(xyz), np.concatenate(desc), str(d)


def test_upload_from_files_equals_upload_from_arrays(model_files):
    paths, xyz, desc, d = model_files
    s = capi.ModelSet()
    for p in paths:
        s.add_xml(p)
    assert np.array_equal(s.xyz, xyz) and np.array_equal(s.desc, desc)
    cont = os.path.join(d, "all.mopeddb")
    s.save(cont)
    t = capi.ModelSet.load(cont)
    base, _, _ = synth.load_sift_fixture()
    q = orclib.normalize(base[:700])      # mh_match takes normalised queries
    model_of = s.model_of
    # reference path: oracle normalisation (A1) on the host, then arrays
    a = capi.Context(0)
    a.db_upload(orclib.normalize(desc), model_of, xyz, 4)
    want = a.match(q.copy(), 0.8)
    for src in (s, t):
        b = capi.Context(0)
        src.upload(b)
        got = b.match(q.copy(), 0.8)
        for w, g in zip(want, got):
            assert np.array_equal(np.asarray(w), np.asarray(g))
        b.close()
    # oracle: exact search over the same rows
    oi, o1, o2 = orclib.match_2nn(orclib.normalize(desc), q)
    assert np.array_equal(want[1], oi) and np.array_equal(want[2], o1) and np.array_equal(want[3], o2)
    assert np.array_equal(want[0], np.where(o1 / o2 < np.float32(0.8), oi, -1))
    a.close()
    s.close()
    t.close()


def test_model_blocks_are_shards(model_files):
    paths, xyz, desc, d = model_files
    s = capi.ModelSet()
    for p in paths:
        s.add_xml(p)
    base, _, _ = synth.load_sift_fixture()
    q = orclib.normalize(base[100:500])
    whole = capi.Context(0)
    s.upload(whole)
    want_idx = whole.match(q.copy(), 0.8)[1]               # raw nearest row of every query
    lo = capi.Context(0)
    hi = capi.Context(0)
    s.upload(lo, 0, 2)
    s.upload(hi, 2, 2)
    i_lo = lo.match(q.copy(), 0.8)[1]
    i_hi = hi.match(q.copy(), 0.8)[1]
    split = s.model_range(2)[0]
    assert (i_lo < split).all() and (i_hi >= split).all()     # global row ids (index_base)
    assert ((want_idx == i_lo) | (want_idx == i_hi)).all()
    for c in (whole, lo, hi):
        c.close()
    s.close()

"""N3: `.moped.xml` parsing and the packed `.mopeddb` container.  Host code only.

Golden: tests/golden/models/*.moped.xml and what the reference's own sXML.hpp + stream
operators read out of them (tests/golden/model_xml_ref.npz, oracle/make_golden.py models)."""
import os

import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = os.path.join(HERE, "golden", "models")
GOLD = np.load(os.path.join(HERE, "golden", "model_xml_ref.npz"))
FILES = [str(f) for f in GOLD["files"]]
REGULAR = [f for f in FILES if not f.startswith("quirks")]


def _gold(fn):
    k = fn.split(".")[0]
    return {f: GOLD[f"{k}_{f}"] for f in ("name", "xyz", "desc", "bbox", "n_bad_len")}


@pytest.mark.parametrize("fn", FILES)
def test_oracle_restatement_reads_what_the_reference_reads(fn):
    g = _gold(fn)
    o = orclib.parse_model_xml(os.path.join(MODELS, fn))
    assert o["name"] == str(g["name"])
    assert np.array_equal(o["xyz"], g["xyz"]) and np.array_equal(o["desc"], g["desc"])
    assert np.array_equal(o["bbox"], g["bbox"]) and o["n_bad_len"] == int(g["n_bad_len"])


@pytest.mark.parametrize("fn", REGULAR)
def test_loader_reads_what_the_reference_reads(fn):
    g = _gold(fn)
    s = capi.ModelSet()
    s.add_xml(os.path.join(MODELS, fn))
    assert s.n_models == 1 and s.name(0) == str(g["name"])
    b, n, bbox = s.model_range(0)
    assert (b, n) == (0, len(g["xyz"])) and s.n_rows == n
    assert np.array_equal(s.xyz, g["xyz"])          # bit for bit: same decimal -> float rounding
    assert np.array_equal(s.desc, g["desc"])
    assert np.array_equal(bbox, g["bbox"])
    s.close()


def test_loader_corner_cases_follow_the_reference():
    """comment before an element, escaped quotes, attribute order, foreign descriptor type,
    <Observation> children, missing coordinates, '+'/exponent forms, a later <Points> replacing an
    earlier one.  The reference files a point with a short descriptor (and later reads past its
    end, MATCH_ANN_CPU.hpp:88-90); the loader refuses such a file instead."""
    g = _gold("quirks.moped.xml")
    text = open(os.path.join(MODELS, "quirks.moped.xml"), "rb").read()
    s = capi.ModelSet()
    with pytest.raises(capi.MhError, match="descriptor"):
        s.add_xml_buffer(text)
    assert s.n_models == 0
    lines = [ln for ln in text.split(b"\n") if b"oops" not in ln]
    good = b"\n".join(lines)
    s.add_xml_buffer(good)
    assert s.name(0) == str(g["name"]) == 'quirk "quoted" model'
    assert np.array_equal(s.xyz, g["xyz"][:2]) and np.array_equal(s.desc, g["desc"][:2])
    tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"quirks_good_{os.getpid()}.xml")
    open(tmp, "wb").write(good)
    o = orclib.parse_model_xml(tmp)          # the restatement, pinned above, on the reduced file
    os.remove(tmp)
    assert np.array_equal(s.model_range(0)[2], o["bbox"]) and np.array_equal(s.xyz, o["xyz"])
    s.close()


def test_set_order_replacement_and_container_roundtrip(tmp_path):
    s = capi.ModelSet()
    for fn in REGULAR:
        s.add_xml(os.path.join(MODELS, fn))
    golds = [_gold(fn) for fn in REGULAR]
    want_xyz = np.concatenate([g["xyz"] for g in golds])
    want_desc = np.concatenate([g["desc"] for g in golds])
    assert s.n_models == 3 and s.n_rows == len(want_xyz)
    assert np.array_equal(s.xyz, want_xyz) and np.array_equal(s.desc, want_desc)   # Update()'s flatten order
    assert np.array_equal(s.model_of, np.concatenate([np.full(len(g["xyz"]), i, np.int32) for i, g in enumerate(golds)]))
    path = str(tmp_path / "db.mopeddb")
    s.save(path)
    t = capi.ModelSet.load(path)
    assert t.n_models == 3 and [t.name(i) for i in range(3)] == [s.name(i) for i in range(3)]
    assert np.array_equal(t.xyz, want_xyz) and np.array_equal(t.desc, want_desc)
    for i in range(3):
        assert t.model_range(i)[:2] == s.model_range(i)[:2]
        assert np.array_equal(t.model_range(i)[2], golds[i]["bbox"])
    assert t.desc.ctypes.data % 4096 == 0         # mapped section, page aligned: handed to the GPU as it lies
    with pytest.raises(capi.MhError):
        t.add_xml(os.path.join(MODELS, REGULAR[0]))    # a mapped container is read-only
    # a model with an existing name replaces it in place (moped.cpp:141-146)
    rng = np.random.default_rng(1)
    xyz = rng.random((7, 3)).astype(np.float32)
    desc = rng.random((7, 128)).astype(np.float32)
    p = str(tmp_path / "again.moped.xml")
    rx, rd = synth.write_model_xml(p, s.name(1), xyz, desc)
    s.add_xml(p)
    assert s.n_models == 3 and s.model_range(1)[:2] == (len(golds[0]["xyz"]), 7)
    assert np.array_equal(s.xyz, np.concatenate([golds[0]["xyz"], rx, golds[2]["xyz"]]))
    assert np.array_equal(s.desc, np.concatenate([golds[0]["desc"], rd, golds[2]["desc"]]))
    s.close()
    t.close()


def test_container_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.mopeddb")
    open(p, "wb").write(b"MOPEDDB1" + b"\0" * 100)
    with pytest.raises(capi.MhError):
        capi.ModelSet.load(p)
    open(p, "wb").write(b"not a container at all" * 300)
    with pytest.raises(capi.MhError):
        capi.ModelSet.load(p)

"""The two match kernels against each other and against the oracle on the SAME input: the VALU kernel
(`match_kernel`, below 1536 queries by default) and the matrix-pipe kernel (`match_mfma_kernel`) must
give the same bits -- indices, d1, d2 -- whatever the query count.  The choice is read once per process
(`MH_MATCH_MFMA`), so each kernel runs in a child process of its own (one at a time) and writes its
results to a file; the parent compares them bit for bit and checks a sample against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
from moped_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from moped_amd import capi
z = np.load(sys.argv[2])
c = capi.Context(0)
c.db_upload(z["dbn"], z["model_of"], z["xyz"], int(z["n_models"]))
out = {}
for name in ("q_small", "q_ragged", "q_big"):
    acc, raw, d1, d2 = c.match(z[name], ratio=0.8)
    out[name + "_acc"], out[name + "_raw"], out[name + "_d1"], out[name + "_d2"] = acc, raw, d1, d2
np.savez(sys.argv[3], **out)
c.close()
"""


@pytest.fixture(scope="module")
def problem(tmp_path_factory):
    db = synth.make_db(6, 3000)                      # 18000 rows: 141 tiles, not a multiple of any split count
    fr = synth.make_frame(db, n_vis=2, seed=5, Q=2600)
    dbn = orclib.normalize(db.desc)
    dbn[7] = dbn[3]                                  # exact duplicate rows: ties go to the lower row
    dbn[9000] = dbn[3]
    q = orclib.normalize(fr.desc)                    # mh_match takes normalised queries (A1 is its own entry point)
    q[11] = dbn[3]                                   # a query that IS a database row (distance 0 after the clamp)
    d = tmp_path_factory.mktemp("mk")
    inp = os.path.join(d, "in.npz")
    np.savez(inp, dbn=dbn, model_of=db.model_of, xyz=db.xyz, n_models=db.n_models,
             q_small=q[:70], q_ragged=q[:1537], q_big=q)
    outs = {}
    for flag in ("0", "1"):
        out = os.path.join(d, f"out{flag}.npz")
        env = dict(os.environ, MH_MATCH_MFMA=flag)
        r = subprocess.run([sys.executable, "-c", CHILD, ROOT, inp, out], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = dict(np.load(out))
    return dbn, {"q_small": q[:70], "q_ragged": q[:1537], "q_big": q}, outs


@pytest.mark.parametrize("name", ["q_small", "q_ragged", "q_big"])
def test_valu_and_mfma_kernels_give_the_same_bits(problem, name):
    _, _, outs = problem
    for part in ("acc", "raw", "d1", "d2"):
        a, b = outs["0"][f"{name}_{part}"], outs["1"][f"{name}_{part}"]
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (name, part)


def test_both_kernels_agree_with_the_oracle(problem):
    dbn, qs, outs = problem
    qn = qs["q_big"]
    pick = np.r_[0:16, 11, 1500:1600, 2590:2600]
    oi, od1, od2 = orclib.match_2nn(dbn, qn[pick])
    for flag in ("0", "1"):
        raw, d1, d2 = outs[flag]["q_big_raw"], outs[flag]["q_big_d1"], outs[flag]["q_big_d2"]
        assert np.array_equal(raw[pick], oi) and np.array_equal(d1[pick], od1) and np.array_equal(d2[pick], od2)
    # the duplicated row (3 = 7 = 9000): the lowest index wins and the second best equals the best
    for flag in ("0", "1"):
        assert outs[flag]["q_big_raw"][11] == 3 and outs[flag]["q_big_d1"][11] == outs[flag]["q_big_d2"][11]
        assert outs[flag]["q_big_d1"][11] < 1e-6

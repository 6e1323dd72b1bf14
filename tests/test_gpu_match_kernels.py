"""The two exact match kernels against each other and against the oracle on the SAME input: the VALU kernel
(`match_kernel`, below 1536 queries by default) and the f32 matrix-pipe kernel (`match_mfma_kernel`) must give the
same bits -- indices, d1, d2 -- whatever the query count.  Each is pinned per context through the C ABI
(mh_match_set_mode 2 = VALU, 3 = matrix pipe; the process-wide MH_MATCH_MFMA switch only exists in experiment builds),
and the launch counters of the library say which kernel actually ran."""
import numpy as np
import pytest

import orclib
from moped_amd import capi, synth

pytestmark = pytest.mark.gpu
MODES = {"0": 2, "1": 3}     # "0" = VALU kernel, "1" = matrix-pipe kernel


@pytest.fixture(scope="module")
def problem():
    db = synth.make_db(6, 3000)                      # 18000 rows: 141 tiles, not a multiple of any split count
    fr = synth.make_frame(db, n_vis=2, seed=5, Q=2600)
    dbn = orclib.normalize(db.desc)
    dbn[7] = dbn[3]                                  # exact duplicate rows: ties go to the lower row
    dbn[9000] = dbn[3]
    q = orclib.normalize(fr.desc)                    # mh_match takes normalised queries (A1 is its own entry point)
    q[11] = dbn[3]                                   # a query that IS a database row (distance 0 after the clamp)
    qs = {"q_small": q[:70], "q_ragged": q[:1537], "q_big": q}
    outs = {}
    for flag, mode in MODES.items():
        c = capi.Context(0)
        c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
        c.match_set_mode(mode)
        out = {}
        for name, qq in qs.items():
            before = c.match_kernel_launches()
            acc, raw, d1, d2 = c.match(qq, ratio=0.8)
            after = c.match_kernel_launches()
            # the pinned kernel ran, the other one and the two-stage path did not
            ran = {k: after[k] - before[k] for k in after}
            assert ran["valu" if mode == 2 else "mfma"] == 1 and ran["mfma" if mode == 2 else "valu"] == 0 and ran["screen"] == 0, (flag, name, ran)
            out[name + "_acc"], out[name + "_raw"], out[name + "_d1"], out[name + "_d2"] = acc, raw, d1, d2
        outs[flag] = out
        c.close()
    return dbn, qs, outs


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

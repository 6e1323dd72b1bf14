"""The scenes the frame stress runs singled out, as regression fixtures (tests/golden/stress_scenes.json, made by
tests/tools/make_stress_fixtures.py): every scene is a generator parameter set + the oracle's outcome on one thread.

  * CPU: the oracle still gives the stored outcome (the oracle is unpinned for CLUSTER / the RANSAC skeleton / FILTER --
    SURVEY.md 8(c) -- so a change of its behaviour on these eight frames must be a decision, not an accident);
  * GPU: what each class says about the DEVICE, with the explanation asserted --
      marginal   (r04: an object only the oracle reported): found now, same models and counts, score within 2%;
      score      (r04 / first half of r05: FILTER2 score 7-15% under the oracle's at a pose inside the bar): within 2% now,
                 and the explanation still holds -- the model has ONE match that lies 4.5-15 px^2 off the planted pose,
                 i.e. on POSE2's threshold, and the reported pose does NOT give way to it;
      sensitive  (a planted object only the device reports): the oracle reports nothing, the device exactly the planted
                 object whose cluster holds more than MinNPtsObject matches within the threshold of the planted pose."""
import json
import os
import sys

import numpy as np
import pytest

import orclib
from moped_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import stress_scene  # noqa: E402

K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY
SCENES = json.load(open(os.path.join(ROOT, "tests", "golden", "stress_scenes.json")))
IDS = [f"{s['class_']}-{s['params']['stress_seed']}-{s['params']['scene']}" for s in SCENES]


def _oracle(s):
    p = s["params"]
    db, fr = stress_scene.build(p)
    idx, d1, d2 = orclib.match_2nn(orclib.normalize(db.desc), orclib.normalize(fr.desc))
    om, op, osc, oc, oinl = orclib.frame_rest_inliers(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=1,
                                                      seed=p["seed"])
    return db, fr, idx, d1, d2, om, op, osc, oc, oinl


@pytest.mark.parametrize("s", SCENES, ids=IDS)
def test_oracle_outcome_of_the_fixture_scenes(s):
    db, fr, idx, d1, d2, om, op, osc, oc, oinl = _oracle(s)
    o = s["oracle"]
    assert [int(c) for c in oc] == o["counts"] and [int(m) for m in om] == o["models"]
    assert np.allclose(osc, o["scores"], atol=2e-3) and [len(i) for i in oinl] == o["inliers"]
    assert [int(v) for v in fr.visible] == s["visible"]


@pytest.mark.gpu
@pytest.mark.parametrize("s", SCENES, ids=IDS)
def test_device_on_the_fixture_scenes(s):
    import torch
    from moped_amd.pipeline import FramePipeline, ShardedDB
    p = s["params"]
    db, fr, idx, d1, d2, om, op, osc, oc, oinl = _oracle(s)
    dev = torch.device("cuda:0")
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=4000)
    pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=p["seed"])
    objs, counts = pipe.fetch(0)
    pipe.close()
    assert counts[0] == oc[0] and counts[1] == oc[1]          # matches and clusters: index-exact stages
    out_q, off = orclib.match_accept(idx, d1, d2, 0.8, db.model_of, db.n_models)

    def planted_e2(m, pose=None):
        q = out_q[off[m]:off[m + 1]]
        pl = fr.poses[list(fr.visible).index(m)] if pose is None else pose
        return q, ((orclib.project(pl, db.xyz[idx[q]], K, CAM0) - fr.uv[q]) ** 2).sum(1)

    def bar(m, pose, ref=None):
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        e = lambda ps: float(np.sqrt(((orclib.project(ps, db.xyz[fr.src_point[rows]], K, CAM0) - fr.uv[rows]) ** 2).sum(1)).mean())
        return e(pose), (e(ref) if ref is not None else None)

    if s["class_"] in ("marginal", "score"):
        assert sorted(objs["model"].tolist()) == sorted(int(m) for m in om)
        assert counts[3] == oc[3]
        for m, po, so in zip(om, op, osc):
            g = objs[objs["model"] == m][0]
            e_g, e_o = bar(m, g["pose"], po)
            assert e_g <= e_o + 1.0 and e_g < 1.0
            assert abs(g["score"] - so) <= 0.02 * so, (int(m), float(g["score"]), float(so))
    if s["class_"] == "marginal":
        # the explanation: model 1's matches are clean (18 within 1 px^2 of the planted pose) but no mean-shift cluster of
        # them holds more than MinNPtsObject (6) within POSE's threshold -- the object exists only through a dragged fit
        q, e2 = planted_e2(1)
        clusters, _ = orclib.meanshift(fr.uv[q])
        good = [int((e2[np.asarray(c)] < 10.0).sum()) for c in clusters]
        assert int((e2 < 1.0).sum()) >= 17 and max(good) == 6 and 1 in objs["model"].tolist()
    if s["class_"] == "score":
        # the explanation: exactly one of the affected model's matches lies on POSE2's threshold under the planted pose
        # (4.5 .. 15 px^2; everything else is clean or far off), and the reported pose leaves it where the oracle's does
        # instead of pulling it in at the clean inliers' cost
        hit = 0
        for m, po in zip(om, op):
            q, e2 = planted_e2(m)
            border = np.nonzero((e2 > 4.5) & (e2 < 16.0))[0]
            if len(border) != 1:
                continue
            hit += 1
            g = objs[objs["model"] == m][0]
            _, e_dev = planted_e2(m, g["pose"])
            _, e_orc = planted_e2(m, po)
            clean = e2 < 1.0
            assert np.sqrt(e_dev[clean]).mean() <= np.sqrt(e_orc[clean]).mean() + 0.05       # r04: 0.57 against 0.33 px
            assert e_dev[border[0]] >= 0.8 * e_orc[border[0]]                                # r04: pulled to 3-4 px^2
        assert hit >= 1
    if s["class_"] == "sensitive":
        assert len(om) == 0 and objs["model"].tolist() == [2] and 2 in s["visible"]
        q, e2 = planted_e2(2)
        clusters, _ = orclib.meanshift(fr.uv[q])
        good = [(int((e2[np.asarray(c)] < 10.0).sum()), len(c)) for c in clusters]
        assert max(g for g, _ in good) == 7 and (7, 44) in good                               # 7 good matches among 44: passes ':204'
        e_g, _ = bar(2, objs[0]["pose"])
        assert e_g < 1.0

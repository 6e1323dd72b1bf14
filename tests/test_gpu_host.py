"""The C++ host side: the STEP plugin classes (moped_amd/host/*.hpp) wired into a
MopedPipeline exactly like src/config.hpp does, driven by moped_hip_test, must give
the same frame result as the oracle pipeline."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
from moped_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "moped_amd", "host")
K, CAM0 = synth.K_DEFAULT, synth.CAM_IDENTITY


def test_plugin_headers_are_gnu98_clean():
    subprocess.check_call(["make", "-s", "-C", HOST, "check98"])


@pytest.mark.gpu
def test_step_plugins_through_pipeline(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import dump_scene
    subprocess.check_call(["make", "-s", "-C", HOST, "moped_hip_test"])
    db = synth.make_db(6, 2000)
    fr = synth.make_frame(db, n_vis=2, seed=5, Q=1500, pts_per_obj=120)
    scene = str(tmp_path / "scene.bin")
    dump_scene.dump(scene, db, fr)
    out = subprocess.check_output([os.path.join(HOST, "moped_hip_test"), scene, "3"], text=True)
    objs, counts = [], None
    for line in out.splitlines():
        w = line.split()
        if w[0] == "OBJ":
            objs.append((int(w[1].replace("model", "")), np.array([float(x) for x in w[5:9] + w[2:5]], np.float32), float(w[9])))
        if w[0] == "MATCHES":
            counts = (int(w[1]), int(w[3]))
    dbn, qn = orclib.normalize(db.desc), orclib.normalize(fr.desc)
    idx, d1, d2 = orclib.match_2nn(dbn, qn)
    om, op, osc, oc = orclib.frame_rest(fr.uv, idx, d1, d2, db.model_of, db.xyz, db.n_models, K, CAM0, n_threads=2, seed=1)
    assert counts == (int(oc[0]), int(oc[1]))          # matches and clusters: index-exact stages
    assert sorted(m for m, _, _ in objs) == sorted(om.tolist())
    for m, pose, score in objs:
        j = list(om).index(m)
        rows = np.nonzero((fr.src_point >= 0) & ~fr.is_outlier)[0]
        rows = rows[db.model_of[fr.src_point[rows]] == m]
        xyz, uv = db.xyz[fr.src_point[rows]], fr.uv[rows]
        e_g = np.sqrt(((orclib.project(pose, xyz, K, CAM0) - uv) ** 2).sum(1)).mean()
        e_o = np.sqrt(((orclib.project(op[j], xyz, K, CAM0) - uv) ** 2).sum(1)).mean()
        assert e_g <= e_o + 1.0
        assert abs(score - osc[j]) <= 0.05 * osc[j]

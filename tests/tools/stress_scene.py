"""The scenes of tests/tools/frame_stress.py by number: the tool draws every scene's parameters from ONE generator in a
fixed order, so scene i of `frame_stress.py N seed` is a function of (i, seed) -- replayed here without touching a GPU.
Used to turn the scenes the stress run reports (marginal / score / spread classes) into regression fixtures
(tests/golden/stress_scenes.json, tests/test_gpu_stress_fixtures.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np


def scene_params(index: int, seed: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    done = 0
    while True:
        n_models, ppm = int(rng.choice([3, 6, 12, 20])), int(rng.choice([600, 1500, 3000]))
        db_seed = int(rng.integers(1 << 30))
        for k in range(6):
            Q = int(rng.choice([300, 900, 2000, 3000, 4000]))
            n_vis = int(rng.integers(0, min(n_models, 8) + 1))
            pts = int(rng.choice([12, 40, 150, 300]))
            pts = min(pts, ppm // 2, Q // max(n_vis, 1))
            frame_seed = int(rng.integers(1 << 30))
            outlier_frac = float(rng.choice([0.0, 0.2, 0.5]))
            frame_key = int(rng.integers(1, 1 << 20))
            if done == index:
                return dict(scene=index, stress_seed=seed, n_models=n_models, ppm=ppm, db_seed=db_seed, Q=Q, n_vis=n_vis, pts=pts,
                            frame_seed=frame_seed, outlier_frac=outlier_frac, seed=frame_key)
            done += 1


def build(p: dict):
    """(db, frame) of a scene's parameters."""
    from moped_amd import synth
    db = synth.make_db(p["n_models"], p["ppm"], seed=p["db_seed"])
    fr = synth.make_frame(db, n_vis=p["n_vis"], seed=p["frame_seed"], Q=p["Q"], pts_per_obj=p["pts"], outlier_frac=p["outlier_frac"])
    return db, fr


if __name__ == "__main__":
    import json
    print(json.dumps([scene_params(int(a)) for a in sys.argv[1:]], indent=1))

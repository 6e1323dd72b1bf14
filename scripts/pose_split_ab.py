"""POSE as one launch (MH_POSE_SPLIT=0) or two (hypotheses, then one-wavefront refines): the same objects, bit for bit?
Needs the experiment build (MH_LIB_PATH=.../libmoped_hip_exp.so).  Runs itself twice as child processes and compares."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from moped_amd import synth, capi
    from moped_amd.pipeline import FramePipeline, ShardedDB
    db = synth.make_db(20, 5000)
    dev = torch.device("cuda:0")
    out = {}
    B = 8
    pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=2, max_queries=B * 3000, batch=B)
    frs = [synth.make_frame(db, n_vis=n, seed=300 + i) for i, n in enumerate((2, 5, 10, 0, 1, 3, 2, 7))]
    for i, fr in enumerate(frs):
        pipe.enqueue(0, torch.from_numpy(fr.desc).to(dev), torch.from_numpy(fr.uv).to(dev), seed=50 + i)
        objs, counts = pipe.fetch(0)
        out[f"a{i}_pose"], out[f"a{i}_model"], out[f"a{i}_score"], out[f"a{i}_counts"] = objs["pose"], objs["model"], objs["score"], counts
    qd = torch.cat([torch.from_numpy(f.desc) for f in frs]).to(dev); uv = torch.cat([torch.from_numpy(f.uv) for f in frs]).to(dev)
    torch.cuda.synchronize()
    pipe.enqueue_batch(1, qd, uv, B, [50 + i for i in range(B)])
    for i, (objs, counts) in enumerate(pipe.fetch_batch(1, B)):
        out[f"b{i}_pose"], out[f"b{i}_model"], out[f"b{i}_score"], out[f"b{i}_counts"] = objs["pose"], objs["model"], objs["score"], counts
    np.savez(sys.argv[2], **out)
    pipe.close()
    sys.exit(0)
import numpy as np, tempfile
res = {}
for mode in ("0", "1"):
    f = os.path.join(tempfile.gettempdir(), f"pose_split_{mode}.npz")
    subprocess.check_call([sys.executable, __file__, "child", f], env=dict(os.environ, MH_POSE_SPLIT=mode))
    res[mode] = dict(np.load(f))
same = all(np.array_equal(res["0"][k].view(np.uint32) if res["0"][k].dtype == np.float32 else res["0"][k],
                          res["1"][k].view(np.uint32) if res["1"][k].dtype == np.float32 else res["1"][k]) for k in res["0"])
nobj = sum(len(res["1"][k]) for k in res["1"] if k.endswith("_model"))
print("one launch vs two launches: objects of 8 frames alone + the same as a batch:", "bit-identical" if same else "DIFFERENT", f"({nobj} objects)")
for k in res["0"]:
    if k.endswith("_model") and not np.array_equal(res["0"][k], res["1"][k]): print(" ", k, res["0"][k], res["1"][k])
sys.exit(0 if same else 1)

"""Phase cycle counts of the pose kernel over one frame (needs a build with EXTRA=-DPOSE_PROF)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import synth, capi
from moped_amd.pipeline import FramePipeline, ShardedDB
db = synth.make_db(20, 5000)
fr = synth.make_frame(db, n_vis=2, seed=0)
dev = torch.device("cuda:0")
pipe = FramePipeline(0, ShardedDB(db.desc, db.xyz, db.model_of, db.n_models), depth=1, max_queries=3000)
q = torch.from_numpy(fr.desc).to(dev); uv = torch.from_numpy(fr.uv).to(dev)
L = capi.load()
out = (C.c_ulonglong * 32)()
for rep in range(3):
    pipe.enqueue(0, q.clone(), uv, seed=rep + 1); pipe.fetch(0); L.mh_debug_pose_prof(out, 1)
names = ["load+distinct", "hypotheses", "argmax", "inlier list", "LM plain", "LM squared", "-", "tasks"]
n = max(out[7], 1)
print(f"tasks {out[14]} (POSE + POSE2), {out[7]} of them refined an object, the others took {out[15] // max(out[14] - out[7], 1)} cycles each; cycles per refined task: " + "  ".join(f"{nm}={v // n}" for nm, v in zip(names[:6], out))
      + f"  LM iterations per task={out[6] / n:.1f}")
it = max(out[6], 1)
print(f"per LM iteration: Jacobian pass={out[8] // it}  27 wave sums={out[9] // it}  attempts={out[12] / it:.2f}  "
      f"per attempt: solve+update={out[10] // max(out[12], 1)}  cost={out[11] // max(out[12], 1)};  first cost per refine (4 per task)={out[13] // (4 * n)}")
pipe.close()
for ph, nm in ((0, "plain"), (1, "squared")):
    e = [out[16 + 4 * ph + k] for k in range(4)]
    print(f"refines on {nm} residuals: {sum(e)} (cap {e[0]}, eight rejections {e[1]}, converged {e[2]}, zero gradient {e[3]}); "
          f"iterations {out[24 + ph]}, attempts {out[26 + ph]}")

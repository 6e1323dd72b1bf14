"""Where do the two-stage MATCH and the exact kernels disagree?  (debugging aid for new screen kernels)
usage: screen_debug.py [models=20] [Q=24000]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import orclib
from moped_amd import capi, synth
models = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 24000
db = synth.make_db(models, 5000)
dbn = orclib.normalize(db.desc)
frs = [synth.make_frame(db, n_vis=2, seed=200 + i, Q=3000) for i in range((Q + 2999) // 3000)]
qn = np.concatenate([orclib.normalize(f.desc) for f in frs])[:Q]
dev = torch.device("cuda:0")
c = capi.Context(0)
c.db_upload(dbn, db.model_of, db.xyz, db.n_models)
c.reserve(Q)
tq = torch.from_numpy(qn).to(dev)
qnorm = torch.from_numpy(orclib.row_norms(qn)).to(dev)
res = {}
for mode in (1, 0):
    out = [torch.empty(Q, dtype=t, device=dev) for t in (torch.int32, torch.float32, torch.float32)]
    c.match_set_mode(mode)
    c.match_stats(reset=True)
    c.match_local_dev(tq.data_ptr(), qnorm.data_ptr(), Q, *[o.data_ptr() for o in out])
    c.synchronize()
    res[mode] = [o.cpu().numpy() for o in out]
    print("mode", mode, c.match_stats())
i1, a1, b1 = res[1]
i0, a0, b0 = res[0]
bad1 = np.nonzero(i1 != i0)[0]
bad2 = np.nonzero((i1 == i0) & (b1.view(np.uint32) != b0.view(np.uint32)))[0]
print(f"{Q} queries: best row differs for {len(bad1)}, second distance differs for {len(bad2)} more")
for name, bad in (("best", bad1), ("second", bad2)):
    if not len(bad):
        continue
    rows = i0[bad]
    print(name, "first few queries", bad[:12].tolist())
    print("  true best row % 128 histogram by 32-row block:", np.bincount((rows % 128) // 32, minlength=4).tolist())
    print("  true best row % 32 by 16-row tile:", np.bincount((rows % 32) // 16, minlength=2).tolist())
    print("  (row % 16) // 4 (quarter):", np.bincount((rows % 16) // 4, minlength=4).tolist())
    print("  row % 4:", np.bincount(rows % 4, minlength=4).tolist())
    print("  query % 32 by 16-query tile:", np.bincount((bad % 32) // 16, minlength=2).tolist(), " query % 128 // 32:", np.bincount((bad % 128) // 32, minlength=4).tolist())
    print("  d1 two-stage vs exact:", list(zip(a1[bad[:6]].tolist(), a0[bad[:6]].tolist())), " d2:", list(zip(b1[bad[:6]].tolist(), b0[bad[:6]].tolist())))
    dup = (a1[bad] == b1[bad]).sum()
    print(f"  two-stage d1 == d2 (a row counted twice?) in {dup} of {len(bad)}")
c.close()

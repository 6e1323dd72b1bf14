"""Phase tick counts (clock64; read them as shares of a key's time: the tick is not calibrated against the kernel's
duration and the build costs 21 more registers) of describe_kernel over one image alone (needs a build with
EXTRA=-DSIFT_PROF; MH_LIB_PATH).
usage: sift_prof.py [textured|bundled]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from moped_amd import capi, synth
which = sys.argv[1] if len(sys.argv) > 1 else "textured"
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sift_ref_frames.npz"))
gray = synth.textured_image(0) if which == "textured" else gold["gray0"]
dev = torch.device("cuda:0")
h, w = gray.shape
img = torch.from_numpy(np.ascontiguousarray(gray)).to(dev)
cap = 4096
c = capi.Context(0)
d, xy, cnt = torch.empty((cap, 128), device=dev), torch.empty((cap, 2), device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
L = capi.load()
out = (C.c_ulonglong * 32)()
for rep in range(3):
    c.sift_dev(img.data_ptr(), w, h, True, d.data_ptr(), xy.data_ptr(), 0, cap, cnt.data_ptr())
    torch.cuda.synchronize()
    L.mh_debug_sift_prof(out, 1)
names = ["setup+rows", "fetch", "sample math", "masks+sizes", "list writes", "barrier wait", "B", "normalise+out"]
for base, who in ((0, "first wavefront (does B)"), (10, "last wavefront")):
    keys, steps = max(out[base + 8], 1), max(out[base + 9], 1)
    tot = sum(out[base + i] for i in range(8))
    print(f"{which}: {who}: keys {keys}, steps per key {steps / keys:.2f}, cycles per key {tot // keys}; per key: "
          + "  ".join(f"{nm}={out[base + i] // keys}" for i, nm in enumerate(names)))
    print("    per step: " + "  ".join(f"{nm}={out[base + i] // steps}" for i, nm in enumerate(names) if 1 <= i <= 6))
c.close()

"""N3 measurement: how fast does a model database get from disk into HBM?
  * XML parse: this loader vs the reference's sXML + addModel walk (oracle/_ref, if present), MB/s
  * `.mopeddb`: map + upload + on-device normalisation, GB/s of descriptor bytes
usage: model_load_bench.py [n_models_xml=4] [n_models_container=200]"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from moped_amd import capi, synth
n_xml = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_bin = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = tempfile.mkdtemp(prefix="mopeddb_")
db = synth.make_db(n_xml, 5000)
paths = []
for m in range(n_xml):
    rows = db.model_of == m
    p = os.path.join(d, f"m{m}.moped.xml")
    synth.write_model_xml(p, f"model{m}", db.xyz[rows], db.desc[rows])
    paths.append(p)
mb = sum(os.path.getsize(p) for p in paths) / 1e6
capi.load()   # library load is not parsing time
t0 = time.perf_counter()
s = capi.ModelSet()
for p in paths: s.add_xml(p)
t_mine = time.perf_counter() - t0
print(f"XML  {n_xml} models, {s.n_rows} points, {mb:.1f} MB: this loader {t_mine*1e3:8.1f} ms = {mb/t_mine:7.1f} MB/s")
try:
    import orclib
    if orclib.ref_available():
        t0 = time.perf_counter()
        for p in paths: orclib.ref_model_xml(p)
        t_ref = time.perf_counter() - t0
        print(f"     reference sXML + addModel walk (oracle/_ref, 1 thread)  {t_ref*1e3:8.1f} ms = {mb/t_ref:7.1f} MB/s  ({t_ref/t_mine:.1f}x slower)")
except Exception as e:
    print("     (reference parser not available here:", e, ")")
s.close()
# container with n_bin models: built from arrays through the XML-free path (save of a set assembled in memory)
big = synth.make_db(n_bin, 5000)
text_mb = big.n * (128 * 9 + 80) / 1e6
cont = os.path.join(d, "big.mopeddb")
t = capi.ModelSet()
# assemble by parsing small XML buffers would take minutes in python: write the container with numpy instead, in the documented layout
import struct
PAGE = 4096
up = lambda x: (x + PAGE - 1) // PAGE * PAGE
names = b"".join(f"model{m}".encode() + b"\0" for m in range(n_bin))
off_models = PAGE; rec = b""; no = 0
for m in range(n_bin):
    rows = np.nonzero(big.model_of == m)[0]
    bb = np.concatenate([big.xyz[rows].min(0), big.xyz[rows].max(0)]).astype(np.float32)
    nm = f"model{m}".encode()
    rec += struct.pack("<QQ6fII", int(rows[0]), len(rows), *bb.tolist(), no, len(nm)); no += len(nm) + 1
off_names = off_models + len(rec); off_xyz = up(off_names + len(names)); off_desc = up(off_xyz + big.n * 12)
hdr = struct.pack("<8sIIIIQQQQQQQ32s", b"MOPEDDB1", 1, 128, n_bin, 0, big.n, off_models, off_names, len(names), off_xyz, off_desc,
                  off_desc + big.n * 512, b"SIFT")
with open(cont, "wb") as f:
    f.write(hdr); f.write(b"\0" * (off_models - len(hdr))); f.write(rec); f.write(names)
    f.write(b"\0" * (off_xyz - off_names - len(names))); f.write(big.xyz.tobytes())
    f.write(b"\0" * (off_desc - off_xyz - big.n * 12)); f.write(big.desc.tobytes())
import torch
if torch.cuda.is_available():
    c = capi.Context(0)
    for rep in range(3):
        t0 = time.perf_counter()
        t = capi.ModelSet.load(cont)
        t1 = time.perf_counter()
        t.upload(c)
        t2 = time.perf_counter()
        gb = t.n_rows * 512 / 1e9
        print(f".mopeddb {n_bin} models, {t.n_rows} rows, {gb:.2f} GB of descriptors: map {1e3*(t1-t0):6.1f} ms, upload+normalise {1e3*(t2-t1):7.1f} ms "
              f"= {gb/(t2-t0):5.2f} GB/s (page cache {'warm' if rep else 'as written'}); the same rows as XML text would be ~{text_mb:.0f} MB "
              f"= {text_mb/ (mb/t_mine) :.1f} s of parsing with this loader")
        t.close()
    c.close()
import shutil; shutil.rmtree(d)

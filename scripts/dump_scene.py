#!/usr/bin/env python3
"""Write a scene file for moped_amd/host/moped_hip_test (format in its header)."""
import struct
import sys

import numpy as np

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moped_amd import synth  # noqa: E402


def dump(path, db, frame, K=synth.K_DEFAULT, cam=synth.CAM_IDENTITY):
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", db.n_models, frame.desc.shape[0]))
        f.write(np.asarray(K, "<f4").tobytes())
        f.write(np.asarray(cam, "<f4").tobytes())
        for m in range(db.n_models):
            rows = np.nonzero(db.model_of == m)[0]
            f.write(struct.pack("<i", len(rows)))
            f.write(db.xyz[rows].astype("<f4").tobytes())
            f.write(db.desc[rows].astype("<f4").tobytes())
        f.write(frame.uv.astype("<f4").tobytes())
        f.write(frame.desc.astype("<f4").tobytes())


def dump_frames(path, db, frames, K=synth.K_DEFAULT, cam=synth.CAM_IDENTITY):
    """Models + several frames of the same query count (moped_hip_bench: the streaming C++ host)."""
    Q = frames[0].desc.shape[0]
    assert all(f.desc.shape[0] == Q for f in frames)
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", db.n_models, Q, len(frames)))
        f.write(np.asarray(K, "<f4").tobytes())
        f.write(np.asarray(cam, "<f4").tobytes())
        for m in range(db.n_models):
            rows = np.nonzero(db.model_of == m)[0]
            f.write(struct.pack("<i", len(rows)))
            f.write(db.xyz[rows].astype("<f4").tobytes())
            f.write(db.desc[rows].astype("<f4").tobytes())
        for fr in frames:
            f.write(fr.uv.astype("<f4").tobytes())
            f.write(fr.desc.astype("<f4").tobytes())


def dump_images(path, db, frame, images, q_image):
    """Frame with several Images (moped_hip_test --images): images = [(is_map, K[4], cam[7]), ...] in
    FrameData::images order, q_image[Q] = list index of every feature's image."""
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", db.n_models, frame.desc.shape[0], len(images)))
        for is_map, K, cam in images:
            f.write(struct.pack("<i", int(is_map)))
            f.write(np.asarray(K, "<f4").tobytes())
            f.write(np.asarray(cam, "<f4").tobytes())
        for m in range(db.n_models):
            rows = np.nonzero(db.model_of == m)[0]
            f.write(struct.pack("<i", len(rows)))
            f.write(db.xyz[rows].astype("<f4").tobytes())
            f.write(db.desc[rows].astype("<f4").tobytes())
        f.write(frame.uv.astype("<f4").tobytes())
        f.write(frame.desc.astype("<f4").tobytes())
        f.write(np.asarray(q_image, "<i4").tobytes())


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else "scene.bin"
    n_models = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    db = synth.make_db(n_models, 5000)
    dump(out, db, synth.make_frame(db, n_vis=2, seed=0))
    print("wrote", out)

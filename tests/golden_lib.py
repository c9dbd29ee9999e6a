"""Loads tests/golden/*.npz (numpy arrays only; no pickles)."""
import json
import os

import numpy as np

import oracle_lib as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(f[:-4] for f in os.listdir(HERE) if f.endswith(".npz"))


def load(name):
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["spheres"] = g["spheres"].view(O.SPHERE_DTYPE)
    g["materials"] = g["materials"].view(O.MATERIAL_DTYPE)
    g["camera"] = g["camera"].view(O.CAMERA_DTYPE)[0]
    for k in ("width", "height", "spp", "frames", "flags"):
        g[k] = int(g[k])
    return g


def checksums():
    with open(os.path.join(HERE, "checksums.json")) as f:
        return json.load(f)

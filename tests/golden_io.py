"""Load / replay the committed regression vectors under tests/golden/."""
import json
import os

import numpy as np

from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    g["counters"] = json.loads(str(g["counters"]))
    g["atlas"] = np.load(os.path.join(ROOT, "tests", "golden", "atlas.npy"))
    buf = np.zeros(int(g["buffer_size"]), dtype=np.uint64)
    buf[int(g["root_index"]):] = g["descriptors_tail"]
    g["descriptors"] = buf
    return g


def render_with_oracle(g, **kw):
    return orc.raycast(width=int(g["width"]), height=int(g["height"]), cam_dir=g["cam_dir"], cam_pos=g["cam_pos"],
                       lights=g["lights"], atlas=g["atlas"], tile_dim=(16, 16), descriptors=g["descriptors"],
                       root_index=int(g["root_index"]), octree_dim=int(g["dim"]), using_octree=int(g["using_octree"]),
                       grid=g["grid"], max_distance=int(g["max_distance"]), viewport=g["viewport"], trig=g["cam_trig"],
                       active_lights=int(g.get("active_lights", 1)), **kw)

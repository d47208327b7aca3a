#!/usr/bin/env python3
"""Render the headline frame on the GPU and save two pictures for humans: the RGBA8 frame and the
per-pixel DDA step count (a depth-like map).  python tools/render_png.py out_prefix [depth]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
import bench

prefix = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/frame"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sc = bench.build_scene(depth)
c = bench.make_caster(sc, 1920, 1080, 0)
assert c.compute(), c.last_error()
img = c.read_image(); hits = c.read_hits()
# the reference's fog term 1 - steps/700 (ray_caster_kernel.cl:716) is negative at these distances, so the raw
# UNORM8 frame is black where steps > 700; show |rgb| rescaled as well
rgba = c.read_image_rgba8()
Image.fromarray(rgba[..., :3]).resize((960, 540)).save(prefix + "_rgba8.png")
mag = np.abs(img[..., :3]); mag = mag / max(1e-9, np.percentile(mag, 99.5))
Image.fromarray((np.clip(mag, 0, 1) * 255).astype(np.uint8)).resize((960, 540)).save(prefix + "_absrgb.png")
steps = hits[..., 6].astype(np.float32); steps /= max(1.0, steps.max())
shadow = (hits[..., 5] & 4) != 0
vis = np.stack([steps, steps * (1 - 0.5 * shadow), steps * (hits[..., 3] == 5)], axis=-1)
Image.fromarray((vis * 255).astype(np.uint8)).resize((960, 540)).save(prefix + "_steps.png")
print("saved", prefix)

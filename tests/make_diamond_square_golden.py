#!/usr/bin/env python3
"""Writes tests/golden/diamond_square_64.npz: the 64 x 64 height field of Map::GenerateHeightBitmap
(/root/reference/src/map/Map.cpp:144-262) as oracle/diamond_square.py restates it (default-seeded mt19937, corner seed 58) --
the double field before :248 and the uint8 heights after it.  NOT a reference output (Map.cpp needs SFML and cannot be built
in this image): a vector of the second implementation, so the product generator and the oracle cannot drift together."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import diamond_square as ds  # noqa: E402

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "diamond_square_64.npz"), field=ds.height_field(64),
                    height=ds.height_bytes(64), corner_seed=np.float64(58.0))

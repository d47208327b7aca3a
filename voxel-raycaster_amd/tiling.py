"""Row tiling of a frame across GPUs: interleaved bands of `band_rows`
image rows; band b belongs to rank b % world.  Pure host logic shared by the C
layer's vrc_set_row_tiling (same rule), bench.py and the gloo tests."""
from __future__ import annotations

import numpy as np


def rows_of_rank(height: int, rank: int, world: int, band_rows: int = 8) -> np.ndarray:
    """Image rows rendered by `rank`."""
    if world < 1 or not (0 <= rank < world) or band_rows < 8 or band_rows % 8:
        raise ValueError("need 0 <= rank < world and band_rows a multiple of 8")
    y = np.arange(height)
    return y[(y // band_rows) % world == rank]


def merge_tiles(frames, height: int, world: int, band_rows: int = 8) -> np.ndarray:
    """Assemble the full frame from per-rank frames (each full-size, only its rows valid)."""
    out = np.empty_like(frames[0])
    for r in range(world):
        rows = rows_of_rank(height, r, world, band_rows)
        out[rows] = frames[r][rows]
    return out

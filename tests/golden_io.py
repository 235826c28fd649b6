"""Loaders for tests/golden/*.npz (captured from the unmodified reference by
oracle/capture_golden.py; data only)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def grid(name):
    """-> (uint8 grid with reference cell values 0/1/2/3, start_cell, target_cell)"""
    z = load("grids")
    R, C = (int(v) for v in z[name + "_shape"])
    occ = np.unpackbits(z[name + "_bits"])[: R * C].reshape(R, C).astype(np.uint8)
    sr, sc, tr, tc = (int(v) for v in z[name + "_st"])
    g = occ.copy()
    g[sr, sc] = 2
    g[tr, tc] = 3
    return g, sr * C + sc, tr * C + tc


def upsample(g, k):
    """G512/G1024 recipe: np.kron of the obstacle mask, S=(0,0), T=(R-1,C-1)."""
    occ = np.kron((g == 1).astype(np.uint8), np.ones((k, k), np.uint8))
    out = occ.copy()
    out[0, 0] = 2
    out[-1, -1] = 3
    return out


def csr_get(off, flat, i):
    return flat[int(off[i]): int(off[i + 1])]

"""How empty is the deposit bit matrix?  (GPU box, repo root: python scripts/probe_tau_density.py [512|1024] [ants])
After each of the first iterations of the bench MAACO the paths are downloaded and counted: non-zero 64-ant words per cell,
and non-zero (word, 64-cell segment) chunks -- what a per-chunk dirty flag could let k_tau_update skip."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maaco-path-planing_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pathfit  # noqa: E402
from pathfit import env  # noqa: E402
from bench import MAACO_MAIN  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else (16384 if G == 512 else 8192)
grid = env.bench_grid(G)
m = pathfit.MAACO(grid, N, 100, seed=1, **MAACO_MAIN)
RC = G * G
for it in range(1, 9):
    m.iterate_dev(it)
    dc, dl = m.walk_bufs()[0], m.walk_bufs()[1]
    L = dl.download()
    cells = dc.download().reshape(N, -1)
    ants = np.repeat(np.arange(N), np.maximum(L, 0))
    flat = np.concatenate([cells[a, :max(L[a], 0)] for a in range(N)])
    w = ants >> 6
    words = np.unique(w.astype(np.int64) * RC + flat).size
    chunks = np.unique(w.astype(np.int64) * (RC >> 6) + (flat >> 6)).size
    nw = (N + 63) // 64
    print(f"iter {it}: ok ants {(L > 0).sum()}, path cells {flat.size}, non-zero words {words} of {nw * RC} ({100.0 * words / (nw * RC):.2f} %), "
          f"dirty 512-B chunks {chunks} of {nw * (RC >> 6)} ({100.0 * chunks / (nw * (RC >> 6)):.1f} %)", flush=True)

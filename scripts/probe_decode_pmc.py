"""Diagnostic: GA-style decode of a few agents (for PMC instruction counts per pop), prints pops."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import pathfit
from pathfit import env
grid = env.bench_grid(512)
eng = pathfit.Engine(grid)
n, Wp = 64, 5
rng = np.random.default_rng(0)
free = np.flatnonzero(grid.reshape(-1) != 1)
sp = pathfit.score_params(0, True, 0.3, 0.8, 1.8, 100.0)
cap = 16 * 1024 + 64
d_cells, d_len, d_st, d_stats = eng.buf((n, cap), np.int32), eng.buf(n, np.int32), eng.buf(n, np.int32), eng.buf((n, 5), np.float64)
d_wp = eng.put(rng.choice(free, (n, Wp)).astype(np.int32).reshape(-1))
for rep in range(2):
    eng.decode_batch(n, Wp, 0, 512 * 512 - 1, cap, d_cells, d_len, d_st, d_wp, None, sp, d_stats)
c = eng.counters()
print("decode: ms", eng.last_kernel_ms(), "pops", c["pops"], "pushes", c["pushes"], "deckeys", c["decrease_keys"])

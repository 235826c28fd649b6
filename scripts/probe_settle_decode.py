"""How the parallel settling engine fares on the GA decode workload of bench.py (ga512): share of searches certified, kernel time
with the engine off / on.   python scripts/probe_settle_decode.py [agents]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd")]
import numpy as np
import pathfit
from pathfit import env
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
grid = env.bench_grid(512)
eng = pathfit.Engine(grid)
rng = np.random.default_rng(0)
free = np.flatnonzero(grid.reshape(-1) != 1)
sp = pathfit.score_params(0, True, 0.3, 0.8, 1.8, 100.0)
cap = 16 * 1024 + 64
d_cells, d_len, d_st, d_stats = eng.buf((n, cap), np.int32), eng.buf(n, np.int32), eng.buf(n, np.int32), eng.buf((n, 5), np.float64)
d_wp = eng.put(rng.choice(free, (n, 5)).astype(np.int32).reshape(-1))
ref = None
for mode in (0, 1, 0, 1):
    eng.set_option("astar_settle", mode)
    eng.decode_batch(n, 5, 0, 512 * 512 - 1, cap, d_cells, d_len, d_st, d_wp, None, sp, d_stats)
    c = eng.counters()
    ln = d_len.download()
    if ref is None:
        ref = (ln.copy(), d_cells.download().copy())
    else:
        assert np.array_equal(ref[0], ln) and all(np.array_equal(ref[1][i, :ln[i]], d_cells.download()[i, :ln[i]]) for i in range(0, n, 97))
    print(f"settle={mode}: kernel {eng.last_kernel_ms():.1f} ms, pops {c['pops']}, settled {c['settled_searches']}, sequential fallbacks "
          f"{c['sequential_searches']}", flush=True)

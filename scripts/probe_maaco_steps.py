"""Diagnostic: wall time of each MAACO iteration (walk + best scan + pheromone update) on G1024 / G512."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd")]
import numpy as np
import pathfit
from pathfit import env
from pathfit.dist import Comm, ShardedMAACO
sys.path.insert(0, ROOT)
import bench
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = {512: 16384, 1024: 8192}[size]
grid = env.bench_grid(size)
eng = pathfit.Engine(grid)
sm = ShardedMAACO(Comm(None), lambda: pathfit.MAACO(grid, n, 100, engine=eng, seed=0, **bench.MAACO_MAIN), n)
for it in range(1, 10):
    t0 = time.perf_counter()
    sm.step(it)
    eng._ck(eng.L.pf_sync(eng.h))
    print(f"iteration {it}: {1e3 * (time.perf_counter() - t0):.2f} ms  best {sm.local.best_path_length_overall:.2f}", flush=True)

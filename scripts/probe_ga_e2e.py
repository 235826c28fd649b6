"""Diagnostic: wall time of GASolver.solve() generations on G512 with the native / the Python genetic operators."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd")]
import numpy as np
import pathfit
from pathfit import env
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
grid = env.bench_grid(512)
eng = pathfit.Engine(grid)
res = {}
for native in (True, False):
    ga = pathfit.GASolver(grid, 4, N, 5, 0.2, 0.8, engine=eng, seed=1, turn_penalty_factor=0.3, safety_penalty_factor=0.8,
                          min_safe_distance=1.8, diagonal_obstacle_penalty_value=100.0)
    ga.native_operators = native
    t0 = time.perf_counter()
    out = ga.solve()
    dt = time.perf_counter() - t0
    res[native] = (out[5], list(ga.convergence_curve))
    print(f"native={native}: solve() {dt:.2f} s for init + 4 generations of {N}; best fitness {out[5]:.4f}", flush=True)
assert res[True] == res[False], "native and Python operators must give the same run"
print("same best fitness and convergence curve")

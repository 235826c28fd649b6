"""Diagnostic: per-iteration time of the 4096-predator G512 MPA across its three phases (9 iterations, 3 per phase)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), ROOT]
import numpy as np
import pathfit
from pathfit import env
import bench
grid = env.bench_grid(512)
m = pathfit.MPA(grid, 4096, 9, seed=0, **bench.MPA_MAIN)
for it in range(1, 10):
    t0 = time.perf_counter()
    m.step(it)
    dt = time.perf_counter() - t0
    c = m.engine.counters()
    print(f"iter {it} (phase {1 if it <= 3 else (2 if it <= 6 else 3)}): {1e3 * dt:.1f} ms, sweep kernel {m.engine.last_kernel_ms():.1f} ms, pops {c['pops']}, pruned {c['pruned_rebuilds']}, best fitness {m.best_fitness_overall:.3f}", flush=True)

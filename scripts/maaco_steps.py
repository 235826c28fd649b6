"""Step statistics of one MAACO walk batch (how far the longest ant is from the average one): python scripts/maaco_steps.py [size] [ants]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd")]
import numpy as np
import pathfit
from pathfit import env
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ants = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
g = env.bench_grid(size)
m = pathfit.MAACO(g, ants, 100, 1.0, 7.0, 0.1, 2.5, 1.0, 0.9, 0.2, 0.9, 0.5, 0.1, seed=11)
for it in (1, 2, 3):
    m.walk_iteration_dev(it)
    c = m.engine.counters()
    dc, dl, dp, dt, ds = m.walk_bufs()
    ln, st = dl.download(), ds.download()
    print(f"it {it}: kernel {m.engine.last_kernel_ms():.3f} ms, steps/ant {c['steps'] / ants:.1f}, ok {np.mean(st == 0):.3f}, dead {np.mean(st == 1):.3f}, "
          f"len avg {ln[ln > 0].mean():.1f} max {ln.max()} p99 {np.percentile(ln[ln > 0], 99):.0f}", flush=True)

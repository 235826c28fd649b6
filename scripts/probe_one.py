"""One A* batch launch (for PMC profiling): python scripts/probe_one.py <k> <n> <variant>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine
k, n, variant = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g0, _, _ = gio.grid("g256")
g = gio.upsample(g0, k) if k > 1 else g0
R, C = g.shape
e = Engine(g)
rnd = np.random.default_rng(1)
free = np.flatnonzero(g.reshape(-1) != 1)
starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
cap = 8 * (R + C)
ds, dt = e.put(starts), e.put(targets)
dc, dl, dst = e.buf((n, cap), np.int32), e.buf(n, np.int32), e.buf(n, np.int32)
e.astar_batch(variant, ds, dt, n, cap, dc, dl, dst)
c = e.counters()
print("ms", e.last_kernel_ms(), "pops", c["pops"], "pushes", c["pushes"], "nbr", c["nbr_examined"], "deckey", c["decrease_keys"])

"""Diagnostic: a search shaped like the long MPA rebuilds (from a cell early on the initial path to the target,
avoiding the prefix), alone on the chip, through the batch kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
paths, st, cnt = e.astar_host(1, [0], [g.size - 1], None, path_cap=8192, want_counters=True)
p0 = paths[0]
print("init path cells", len(p0), "pops", cnt[0, 0])
for idx in (3, 40, 200):
    cur = int(p0[idx]); avoid = [p0[:idx]]
    for rep in range(2):
        paths, st, cnt = e.astar_host(1, [cur], [g.size - 1], avoid, path_cap=8192, want_counters=True)
    print(f"idx {idx}: {e.last_kernel_ms():.2f} ms pops {cnt[0, 0]} us/pop {1e3 * e.last_kernel_ms() / cnt[0, 0]:.3f}")

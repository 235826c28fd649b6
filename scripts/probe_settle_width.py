"""Diagnostic: the parallel label-settling engine at 1 / 64 / 512 concurrent G512 searches (PF_LIB selects the build: -DPF_ST_WIDE=k)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
rnd = np.random.default_rng(1)
free = np.flatnonzero(g.reshape(-1) != 1)
e.set_option("astar_settle", 1)
for n in (1, 64, 512, 2048):
    for variant in (0, 2):
        starts = np.zeros(n, np.int32); targets = np.full(n, g.size - 1, np.int32)
        if n > 1:
            starts[1:] = rnd.choice(free, n - 1); targets[1:] = rnd.choice(free, n - 1)
        for rep in range(2):
            paths, st, cnt = e.astar_host(variant, starts, targets, None, path_cap=16384, want_counters=True)
        c = e.counters()
        print(f"n={n:5d} v{variant}: {e.last_kernel_ms():8.2f} ms  expansions {int(cnt[:, 0].sum()):10d} ({1e6 * e.last_kernel_ms() / max(cnt[:, 0].sum(), 1):7.2f} ns each)  "
              f"settled {c['settled_searches']} seq {c['sequential_searches']}", flush=True)

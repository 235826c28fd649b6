"""Diagnostic: the closed-set connector on ONE search (G512 / G1024 corner to corner) and on n concurrent copies, sequential
pop loop vs the parallel label-settling engine (pf_settle.h).  python scripts/probe_settle.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine
for k in (2, 4):
    g = gio.upsample(gio.grid("g256")[0], k)
    e = Engine(g)
    rnd = np.random.default_rng(1)
    free = np.flatnonzero(g.reshape(-1) != 1)
    for n in (1, 256, 2048):
        for mode in (0, 1):
            e.set_option("astar_settle", mode)
            for variant in (0, 2):
                starts = np.zeros(n, np.int32); targets = np.full(n, g.size - 1, np.int32)
                if n > 1:
                    starts[1:] = rnd.choice(free, n - 1); targets[1:] = rnd.choice(free, n - 1)
                for rep in range(2):
                    paths, st, cnt = e.astar_host(variant, starts, targets, None, path_cap=16384, want_counters=True)
                c = e.counters()
                print(f"G{256 * k} n={n:5d} {'settle' if mode else 'seq   '} v{variant}: {e.last_kernel_ms():8.2f} ms  expansions {int(cnt[:, 0].sum()):10d} "
                      f"({1e6 * e.last_kernel_ms() / max(cnt[:, 0].sum(), 1):8.2f} ns each)  settled {c['settled_searches']} seq {c['sequential_searches']}", flush=True)
    e.close()

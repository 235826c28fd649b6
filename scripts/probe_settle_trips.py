"""Diagnostic: nodes per trip of the parallel label-settling engine on A* (v0) and Dijkstra (v2) searches of G512 (random pairs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
rnd = np.random.default_rng(3)
free = np.flatnonzero(g.reshape(-1) != 1)
e.set_option("astar_settle", 1)
n = 64
starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
for variant in (0, 2):
    paths, st, cnt = e.astar_host(variant, starts, targets, None, path_cap=16384, want_counters=True)
    c = e.counters()
    ok = cnt[:, 2] > 0
    npt = cnt[ok, 0] / np.maximum(cnt[ok, 2], 1)
    print(f"v{variant}: {e.last_kernel_ms():.2f} ms, settled {c['settled_searches']} seq {c['sequential_searches']}; expansions/trip p10 {np.percentile(npt, 10):.1f} p50 {np.percentile(npt, 50):.1f} p90 {np.percentile(npt, 90):.1f}; "
          f"trips p50 {np.percentile(cnt[ok, 2], 50):.0f} max {cnt[ok, 2].max()}; expansions p50 {np.percentile(cnt[ok, 0], 50):.0f} max {cnt[:, 0].max()}")

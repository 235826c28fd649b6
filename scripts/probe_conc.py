"""Diagnostic: speed of ONE search (G512 corner to corner) when n identical copies run concurrently."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit import _lib
if os.environ.get("PF_LIB"):
    _lib._SO = os.path.abspath(os.environ["PF_LIB"])
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
for n in (1, 64, 256, 512, 1024, 1536, 1792):
    starts = np.zeros(n, np.int32); targets = np.full(n, g.size - 1, np.int32)
    for v in (1,):
        for rep in range(2):
            paths, st, cnt = e.astar_host(v, starts, targets, None, path_cap=8192, want_counters=True)
        print(f"n={n:5d} v{v}: {e.last_kernel_ms():7.2f} ms  us/pop {1e3 * e.last_kernel_ms() / cnt[0, 0]:.3f}  aggregate {cnt[:, 0].sum() / e.last_kernel_ms() / 1e3:8.1f} Mpops/s", flush=True)

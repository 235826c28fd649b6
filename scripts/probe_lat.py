"""Per-pop latency probe: kernel time / max pops of any agent, few agents (no contention)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine

g0, _, _ = gio.grid("g256")
for k in [int(a) for a in sys.argv[1:]] or (1, 2, 4):
    g = gio.upsample(g0, k) if k > 1 else g0
    R, C = g.shape
    e = Engine(g)
    rnd = np.random.default_rng(1)
    free = np.flatnonzero(g.reshape(-1) != 1)
    for n in (32, 256, 1792, 7168):
        starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
        starts[0], targets[0] = 0, R * C - 1
        cap = 8 * (R + C)
        ds, dt = e.put(starts), e.put(targets)
        dc, dl, dst, dcnt = e.buf((n, cap), np.int32), e.buf(n, np.int32), e.buf(n, np.int32), e.buf((n, 4), np.int64)
        for variant in (0, 1):
            e.astar_batch(variant, ds, dt, n, cap, dc, dl, dst, d_counters=dcnt)
            ms = e.last_kernel_ms(); cnt = dcnt.download(); c = e.counters()
            mx = cnt[:, 0].max(); tot = cnt[:, 0].sum()
            print(f"grid {R} n={n} v{variant}: {ms:.1f} ms, max pops {mx}, sum {tot}, >= {ms*1e3/mx:.2f} us/pop (single wave bound), "
                  f"max_open {cnt[:,2].max()}, {tot/ms/1e3:.1f} Mpops/s, ovf {c['overflow_agents']}", flush=True)
    e.close()

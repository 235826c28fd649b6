"""Randomised parity campaign: HIP A* (variants 0, 1, 2) vs the CPU oracle on several maps, paths and pop counts.
    python scripts/soak_parity.py [pairs_per_map] [settle]
With `settle` the closed-set variants (0, 2) go through the parallel label-settling engine first (pf_settle.h): paths and
statuses are compared (its expansion counts are its own) and the share it certified is printed per map."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
if os.environ.get("PF_LIB"):
    from pathfit import _lib
    _lib._SO = os.path.abspath(os.environ["PF_LIB"])
from pathfit.engine import Engine
from pathfit import env
import golden_io as gio
import pf_oracle as po
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
SETTLE = len(sys.argv) > 2 and sys.argv[2] == "settle"
rnd = np.random.default_rng(2024)
maps = {"G512": gio.upsample(gio.grid("g256")[0], 2), "G256": gio.grid("g256")[0], "blocks384": env.random_blocks(384, 384, 0.25, seed=11, block=(2, 9)),
        "sparse300": (rnd.random((300, 300)) < 0.08).astype(np.uint8), "empty200": np.zeros((200, 200), np.uint8)}
bad = 0
t00 = time.time()
for name, g in maps.items():
    e, o = Engine(g), po.Oracle(g)
    e.set_option("astar_settle", 1 if SETTLE else 0)
    free = np.flatnonzero(g.reshape(-1) != 1)
    starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
    avoid = []
    for i in range(n):
        k = i % 4
        if k == 0: avoid.append(None)
        elif k == 1: avoid.append(rnd.choice(free, int(rnd.integers(1, 400))))
        elif k == 2:                                   # a path-like avoid set (as MPA/GA build them)
            p, _ = o.astar(int(starts[i]), int(rnd.choice(free)), None, 1)
            avoid.append(p[:-1] if len(p) > 1 else None)
        else: avoid.append(rnd.choice(free, 30))
    for v in (0, 1, 2):
        t0 = time.time()
        paths, st, cnt = e.astar_host(v, starts, targets, avoid, path_cap=min(g.size, 16 * sum(g.shape)), want_counters=True)
        mism = 0
        for i in range(n):
            want, ost = o.astar(int(starts[i]), int(targets[i]), avoid[i], v)
            counts_ok = (SETTLE and v != 1) or len(want) <= 1 or cnt[i, 0] == ost[0]
            if st[i] == 3 or not np.array_equal(paths[i], want) or not counts_ok or (st[i] == 0) != (ost[5] == 0):
                mism += 1
        bad += mism
        c = e.counters()
        print(f"{name} v{v}: {n} searches, {int(cnt[:, 0].sum())} pops, mismatches {mism}, settled {c['settled_searches']} / sequential "
              f"{c['sequential_searches']}, {time.time() - t0:.1f} s", flush=True)
    e.close()
print("TOTAL mismatches", bad, f"{time.time() - t00:.0f} s")
sys.exit(1 if bad else 0)

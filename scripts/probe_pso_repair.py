"""How many particles of a PSO repair round decode the same rounded waypoints as in the round before?  (GPU box)
    python scripts/probe_pso_repair.py [sweeps]
Wraps Engine.decode_raw of the bench's asynchronous pso512 swarm: per launch, the particles evaluated, how many of them were
evaluated earlier in the same sweep, and how many of those round (half-even, pso.py:61-70) to the same W cells as then."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maaco-path-planing_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pathfit  # noqa: E402
from pathfit import env  # noqa: E402
from bench import W_MAIN  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
grid = env.bench_grid(512)
eng = pathfit.Engine(grid)
ps = pathfit.PSOSolver(grid, num_iterations=K, num_particles=2048, num_waypoints_per_particle=5, w=0.7, c1=1.5, c2=1.5, engine=eng, seed=1,
                       asynchronous=True, **W_MAIN)
assert ps.begin()
W = 5
seen = {}
orig = eng.decode_raw
base = {}


def wrapped(m, W_, s, t, cap, cells, ln, st, pos, *rest, **kw):
    d = ps._d
    l0 = (pos - d["pos"].ptr) // (W * 2 * 8)
    P = eng.read(pos, m * W * 2, np.float64).reshape(m, W, 2)
    rc = np.rint(P).astype(np.int64)                      # numpy rint = half to even, like Python's round()
    again = same = 0
    for i in range(m):
        key = l0 + i
        if key in seen:
            again += 1
            same += int(np.array_equal(seen[key], rc[i]))
        seen[key] = rc[i]
    print(f"   launch: particles [{l0}, {l0 + m}) evaluated; {again} of them again, {same} with unchanged rounded waypoints", flush=True)
    return orig(m, W_, s, t, cap, cells, ln, st, pos, *rest, **kw)


eng.decode_raw = wrapped
for k in range(K):
    seen.clear()
    print(f"sweep {k + 1}:")
    ps.sweep()

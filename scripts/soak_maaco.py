"""Randomised MAACO campaign: pathfit.MAACO (GPU) against the oracle-driven loop (pf_loops.maaco_solve) on random maps, colony sizes,
parameters and iteration counts -- best path, length, turns, convergence curve and the pheromone matrix, bit for bit.
    python scripts/soak_maaco.py [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import pathfit  # noqa: E402
from pathfit import env  # noqa: E402
import pf_oracle as po  # noqa: E402
import pf_loops  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = np.random.default_rng(77)
bad = 0
t0 = time.time()
for case in range(n_cases):
    R, C = int(rnd.integers(12, 70)), int(rnd.integers(12, 70))
    g = env.random_blocks(R, C, float(rnd.uniform(0.05, 0.3)), seed=int(rnd.integers(1 << 30)), block=(1, 5)).astype(int)
    free = np.argwhere(g != 1)
    s, t = free[0], free[-1]
    g[tuple(s)], g[tuple(t)] = 2, 3
    ants = int(rnd.choice([1, 7, 33, 64, 65, 130, 500, 1500]))
    iters = int(rnd.integers(2, 7))
    kw = dict(alpha=float(rnd.choice([1.0, 1.0, 1.5])), beta=float(rnd.choice([2.0, 5.0, 7.0])), rho=float(rnd.uniform(0.05, 0.4)), Q=2.5,
              a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=float(rnd.uniform(0.1, 0.9)))
    seed = int(rnd.integers(1 << 20))
    m = pathfit.MAACO(g, ants, iters, C0_initial_pheromone=0.1, seed=seed, **kw)
    path, length, turns = m.solve_path_planning()
    ref = pf_loops.maaco_solve(po.Oracle(g), int(s[0]) * C + int(s[1]), int(t[0]) * C + int(t[1]), ants, iters, C0=0.1, seed=seed, **kw)
    ok = ([r * C + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"]
          and m.convergence_curve_data == ref["curve"] and np.array_equal(m.pheromone_matrix, ref["tau"]))
    bad += not ok
    print(f"case {case}: {R}x{C}, {ants} ants, {iters} iterations, alpha {kw['alpha']}, beta {kw['beta']}: {'ok' if ok else 'MISMATCH'} (best {length})", flush=True)
print("TOTAL mismatches", bad, f"{time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

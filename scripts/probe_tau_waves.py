"""Which wavefronts of k_tau_update are the slow ones?  Needs a -DPF_TAU_PROBE=2 build (PF_LIB=...): the kernel then leaves, per
64-cell stretch, its shader clocks / dirty chunks / dense chunks in tau instead of the pheromone.
    PF_LIB=$PWD/maaco-path-planing_amd/lib/libpf_probe2.so python scripts/probe_tau_waves.py [512|1024]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maaco-path-planing_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pathfit  # noqa: E402
from pathfit import env  # noqa: E402
from bench import MAACO_MAIN  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else (16384 if G == 512 else 8192)
grid = env.bench_grid(G)
m = pathfit.MAACO(grid, N, 100, seed=1, **MAACO_MAIN)
m.iterate_dev(1)
tau = np.asarray(m.pheromone_matrix).reshape(-1)
nseg = tau.size // 64
t = tau[: nseg * 64].reshape(nseg, 64)
clk, chunks, dense = t[:, 0], t[:, 1], t[:, 2]
order = np.argsort(-clk)
print(f"{nseg} stretches; clocks: max {clk.max():.0f}, p99 {np.percentile(clk, 99):.0f}, p90 {np.percentile(clk, 90):.0f}, median {np.median(clk):.0f}; "
      f"sum {clk.sum():.3g} (= {clk.sum() / 1024 / 2.0e6:.3f} ms if spread over 1024 SIMDs at 2 GHz)")
for s in order[:12]:
    print(f"  stretch {s} (row {s * 64 // G}, col {s * 64 % G}): {clk[s]:.0f} clocks, {chunks[s]:.0f} dirty chunks, {dense[s]:.0f} dense; clocks per chunk {clk[s] / max(chunks[s], 1):.0f}")
sel = dense > 0
print(f"stretches with dense chunks: {sel.sum()}; clocks per chunk there {clk[sel].sum() / max(chunks[sel].sum(), 1):.0f}; elsewhere {clk[~sel].sum() / max(chunks[~sel].sum(), 1):.0f}")

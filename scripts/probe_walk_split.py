"""Where a wavefront of k_maaco_walk8 spends its clocks (needs a -DPF_WALK_PROBE build: PF_LIB=...):
    PF_LIB=$PWD/maaco-path-planing_amd/lib/libpf_walkprobe.so python scripts/probe_walk_split.py [512|1024] [ants]
per lock-step round of eight ants: clocks waiting for the round's three loads, clocks marking finished paths, the rest."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maaco-path-planing_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pathfit  # noqa: E402
from pathfit import env  # noqa: E402
from bench import MAACO_MAIN  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else (16384 if G == 512 else 8192)
m = pathfit.MAACO(env.bench_grid(G), N, 100, seed=1, **MAACO_MAIN)
for it in (1, 2, 3):
    m.walk_iteration_dev(it)
    c = m.engine.counters()
    waves = (N + 7) // 8
    rounds, total, wait, mark, longest = c["nbr_examined"], c["pushes"], c["pops"], c["decrease_keys"], c["pruned_rebuilds"]
    print(f"it {it}: kernel {m.engine.last_kernel_ms():.3f} ms; {waves} wavefronts, {rounds / waves:.0f} rounds each, longest wavefront {longest} clocks; "
          f"per round: {total / rounds:.0f} clocks = wait for the loads {wait / rounds:.0f} + marking {mark / rounds:.0f} + rest {(total - wait - mark) / rounds:.0f}", flush=True)

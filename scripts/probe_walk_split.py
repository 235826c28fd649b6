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
    act = max(c["path_cells"], 1)          # rounds in which group 0 of a wavefront stepped: the in-step stamps are lane 0's
    sel0, sel1, head, upd, loop = c["settled_searches"], c["sequential_searches"], c["candidates"], c["steps"], c["overflow_agents"]
    step = (head + sel0 + sel1 + upd) / act
    print(f"it {it}: kernel {m.engine.last_kernel_ms():.3f} ms; {waves} wavefronts, {rounds / waves:.0f} rounds each, longest wavefront {longest} clocks, {total / rounds:.0f} clocks per round")
    print(f"   a step ({step:.0f}): cell / addresses / move mask {head / act:.0f}; issue of the three loads + wait {wait / act:.0f}; candidate masks, RNG, q {(sel0 - wait) / act:.0f}; "
          f"selection (greedy and roulette branches) {sel1 / act:.0f}; move, tabu word, path store {upd / act:.0f}")
    print(f"   around it, per round: marking of finished paths {mark / rounds:.0f}; loop test + fetch test {loop / rounds:.0f}; emit + stamps {total / rounds - step - (mark + loop) / rounds:.0f}", flush=True)

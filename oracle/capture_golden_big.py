"""Golden vectors from the UNMODIFIED reference at the bench sizes -> tests/golden/big_cases.npz.

    python oracle/capture_golden_big.py            (build container only: needs /root/reference; ~10 minutes of CPU)

The grids are the bench grids (pathfit.env.bench_grid: np.kron 2x / 4x of the reference's 256x256 map, env.py:114), rebuilt
here from the reference's own array and compared by hash.  Captured: AStarSolver.solve (astar.py:33-101) and MPA._a_star
(MPA.py:106-151) on pairs of G512 and G1024 -- nearby pairs, path-prefix avoid sets as MPA builds them, and the corner-to-corner
search of each grid -- with paths and heap pop / push counts; GASolver._reconstruct_path_from_chromosome + stats (ga_solver.py:58-93)
with 3 and 5 waypoints on G512; MPA._reconstruct_path_segment (MPA.py:284-318) of the reference's own initial path on G512;
MAACO._construct_ant_solution_maaco + _update_pheromone_trails_maaco (MAACO.py:278-332) on G512 (8 ants x 2 iterations, beta 7 and
beta 2) and G1024 (4 ants x 1 iteration) with the per-agent-keyed streams: every ant's path / length / turns and the pheromone
matrix after each update.  Data only (inputs and the reference's outputs), never reference source text.

    python oracle/capture_golden_big.py maaco      (re-capture only the MAACO keys, ~2 minutes; the other keys are kept)
"""
import os
import platform
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "maaco-path-planing_amd"))
import ref_harness as rh  # noqa: E402
from pathfit import env as pfenv  # noqa: E402
from capture_golden import csr, MAIN_W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
META = dict(python=platform.python_version(), numpy=np.__version__)


def big_grid(k):
    """kron-k upsample of the reference's 256 x 256 map, start (0,0), target (R-1,C-1): must BE pathfit.env.bench_grid(256 k)."""
    g256 = np.array(rh.mods()["env"].grid_map_from_image_data5)
    mask = (g256 == 1).astype(int)
    g = np.kron(mask, np.ones((k, k), int))
    g[0, 0] = 2
    g[-1, -1] = 3
    mine = pfenv.bench_grid(256 * k)
    assert np.array_equal(g == 1, np.asarray(mine) == 1) and pfenv.grid_hash(g) == pfenv.grid_hash(mine)
    return g


def harness_grid(g):
    """The same obstacles with the start / target markers next to each other, so that constructing the reference's MPA (which
    runs one full A* for its initial population, MPA.py:231-245) costs nothing; markers are free cells either way."""
    h = np.array(g).copy()
    h[h > 1] = 0
    free = np.argwhere(h == 0)
    a = tuple(free[0])
    for b in ((a[0], a[1] + 1), (a[0] + 1, a[1])):
        if h[b] == 0:
            h[a] = 2
            h[b] = 3
            return h
    raise RuntimeError("no adjacent free pair")


def cap_maaco_big():
    """MAACO walks and pheromone updates of the unmodified reference at the bench sizes (main.py:34-38 parameters).  The
    pheromone matrices are stored in full: after the first update nearly every free cell sits on a clip bound (MAACO.py:317-331:
    tau_max = 1 / (0.9 L_best) is far below the initial C0 * d(S,T) / (d(S,i) + d(i,T))), so they compress to a few KB."""
    base = dict(alpha=1.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9,
                q0_initial=0.5, C0_initial_pheromone=0.1)
    d = {}
    plan = [(2, 7.0, 8, 2, 100, 2001), (2, 2.0, 8, 2, 100, 2002), (4, 7.0, 4, 1, 100, 2003)]
    for ri, (k, beta, n_ants, n_it, K, seed) in enumerate(plan):
        g = big_grid(k)
        R, C = g.shape
        t0 = time.time()
        ma = rh.make_maaco(g, num_ants=n_ants, num_iterations=K, beta=beta, **base)
        paths, lens, turns, draws, taus = [], [], [], [], []
        best = float("inf")
        for it in range(1, n_it + 1):
            it_paths = []
            for ant in range(n_ants):
                pc, L, T, nd = rh.maaco_walk(ma, it, seed, ant)
                paths.append(pc); lens.append(L); turns.append(T if T != float("inf") else -1); draws.append(nd)
                it_paths.append((rh.to_rc(pc, C), L, T))
                best = min(best, L)
            ma.best_path_length_overall = best                    # MAACO.py:351-352 precedes :359
            ma._update_pheromone_trails_maaco(it_paths, None)
            taus.append(ma.pheromone_matrix.copy())
        po_, pf = csr(paths)
        d[f"maaco{ri}_path_off"], d[f"maaco{ri}_path"] = po_, pf
        d[f"maaco{ri}_len"] = np.array(lens); d[f"maaco{ri}_turns"] = np.array(turns); d[f"maaco{ri}_draws"] = np.array(draws)
        d[f"maaco{ri}_tau"] = np.array(taus)
        d[f"maaco{ri}_cfg"] = np.array([R, beta, n_ants, n_it, K, seed], np.float64)
        print(f"maaco big run {ri}: G{R} beta {beta}: {sum(np.isfinite(lens))} / {len(lens)} ants arrive, longest walk "
              f"{max(len(p) for p in paths)} cells, {time.time() - t0:.0f} s", flush=True)
    d["maaco_base_params"] = np.array([base[k] for k in ("alpha", "rho", "Q", "a_turn_coef", "wh_max", "wh_min",
                                                         "k_h_adaptive", "q0_initial", "C0_initial_pheromone")])
    d["maaco_runs"] = np.array(len(plan))
    return d


def main():
    if sys.argv[1:] == ["maaco"]:                                 # only the MAACO keys: the rest of the file is kept as captured
        z = np.load(os.path.join(OUT, "big_cases.npz"), allow_pickle=False)
        keep = {k: z[k] for k in z.files if not k.startswith("maaco")}
        keep.update(cap_maaco_big())
        np.savez_compressed(os.path.join(OUT, "big_cases.npz"), **keep)
        return
    rnd = random.Random(4242)
    rows, dec = [], []
    t00 = time.time()
    for k, npairs, rad in ((2, 36, 220), (4, 12, 260)):
        g = big_grid(k)
        R, C = g.shape
        name = f"G{R}"
        ra = rh.RefAStar(g)
        mpa = rh.make_mpa(harness_grid(g), num_predators=1)
        free = [tuple(int(v) for v in x) for x in np.argwhere(g != 1)]
        plan = []
        for t in range(npairs):
            s = rnd.choice(free)
            near = [f for f in free if abs(f[0] - s[0]) <= rad and abs(f[1] - s[1]) <= rad]
            e = rnd.choice(near)
            plan.append((s, e, None))
        # path-prefix avoid sets (what MPA._reconstruct_path_segment and the GA / PSO decodes pass): search from the middle of a
        # captured path to its goal with the first half avoided
        p0, _, _ = ra.solve(plan[0][0], plan[0][1], None)
        if len(p0) > 8:
            cells = rh.to_rc(p0, C)
            plan.append((cells[len(cells) // 2], cells[-1], cells[:len(cells) // 2]))
        plan.append(((0, 0), (R - 1, C - 1), None))              # corner to corner: the searches the sweeps end on
        for s, e, avoid in plan:
            for variant in (0, 1):
                if R > 512 and variant == 1 and (s, e) == ((0, 0), (R - 1, C - 1)):
                    continue                                     # (MPA._a_star's linear open-list scans at 1024^2: ~ten minutes)
                t0 = time.time()
                if variant == 0:
                    pc, _, cnt = ra.solve(s, e, avoid)
                else:
                    pc, _, cnt = rh.mpa_astar(mpa, s, e, avoid)
                rows.append(dict(grid=name, variant=variant, start=s[0] * C + s[1], target=e[0] * C + e[1],
                                 avoid=[a[0] * C + a[1] for a in avoid] if avoid is not None else [], has_avoid=avoid is not None,
                                 path=pc, pops=cnt["pops"], pushes=cnt["pushes"]))
                print(f"{name} v{variant} {s}->{e}: {len(pc)} cells, {cnt['pops']} pops, {time.time() - t0:.1f} s", flush=True)
        if R == 512:
            for t in range(5):
                W = 3 if t < 2 else 5
                ga = rh.make_ga(harness_grid(g), W=W, **MAIN_W)
                ga.start_node, ga.target_node = (0, 0), (R - 1, C - 1)     # (attributes astar.py:21-22 set from the markers)
                near = [f for f in free if f[0] < 200 and f[1] < 200] if t < 2 else free
                chrom = [rnd.choice(near) for _ in range(W)]
                t0 = time.time()
                with rh.quiet():
                    p = ga._reconstruct_path_from_chromosome(chrom)
                    stats = ga._calculate_stats_for_path(p)
                dec.append(dict(wp=np.array([c[0] * C + c[1] for c in chrom] + [-1] * (5 - W), np.int32), path=rh.to_cells(p, C),
                                stats=[stats[1], stats[2], stats[3], stats[4], stats[5]]))
                print(f"{name} decode {chrom}: {len(p)} cells, fitness {stats[5]:.3f}, {time.time() - t0:.1f} s", flush=True)
    # MPA._reconstruct_path_segment (MPA.py:284-318) at 512^2, main.py:44-52 parameters: the predator's path is the reference's
    # own initial path (MPA.py:154, 231-245), rebuilt from early, middle and late indices, Brownian and Levy proposals
    g = big_grid(2)
    R, C = g.shape
    mpa = rh.make_mpa(g, num_predators=1, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8, min_safe_distance=1.8,
                      diagonal_obstacle_penalty=100.0)
    base = rh.to_cells(mpa.population[0]["path"], C)
    reb = []
    for t, (idx, is_levy, scale) in enumerate(((3, False, 0.5), (40, False, 0.5), (200, False, 0.5), (len(base) - 30, False, 0.5),
                                               (120, True, 0.5), (350, False, 0.25), (10, True, 0.05), (500, False, 0.5))):
        t0 = time.time()
        pc, res, draws = rh.mpa_rebuild(mpa, rh.to_rc(base, C), rh.to_rc(base, C), idx, is_levy, scale, 777, 3, t)
        reb.append(dict(idx=idx, is_levy=int(is_levy), scale=scale, agent=t, out=pc, draws=draws, stats=[res[1], res[2], res[3], res[4], res[5]]))
        print(f"G512 rebuild idx {idx} levy {is_levy}: {len(pc)} cells, changed {not np.array_equal(pc, base)}, fitness {res[5]:.3f}, {time.time() - t0:.1f} s", flush=True)
    ro, rf = csr([r["out"] for r in reb])
    names = sorted(set(r["grid"] for r in rows))
    ao, af = csr([r["avoid"] for r in rows])
    po_, pf = csr([r["path"] for r in rows])
    dpo, dpf = csr([d["path"] for d in dec])
    np.savez_compressed(
        os.path.join(OUT, "big_cases.npz"), grid_names=np.array(names),
        grid_id=np.array([names.index(r["grid"]) for r in rows]), variant=np.array([r["variant"] for r in rows]),
        start=np.array([r["start"] for r in rows]), target=np.array([r["target"] for r in rows]),
        has_avoid=np.array([r["has_avoid"] for r in rows]), avoid_off=ao, avoid=af, path_off=po_, path=pf,
        pops=np.array([r["pops"] for r in rows]), pushes=np.array([r["pushes"] for r in rows]),
        dec_wp=np.array([d["wp"] for d in dec]), dec_path_off=dpo, dec_path=dpf, dec_stats=np.array([d["stats"] for d in dec]),
        reb_base=base, reb_base_stats=np.array([mpa.population[0][k] for k in ("length", "turns", "safety_penalty", "diag_penalty", "fitness")], np.float64),
        reb_idx=np.array([r["idx"] for r in reb]), reb_is_levy=np.array([r["is_levy"] for r in reb]),
        reb_scale=np.array([r["scale"] for r in reb]), reb_agent=np.array([r["agent"] for r in reb]), reb_draws=np.array([r["draws"] for r in reb]),
        reb_out_off=ro, reb_out=rf, reb_stats=np.array([r["stats"] for r in reb], np.float64), reb_seed_it=np.array([777, 3]),
        reb_sigma=np.array([rh.levy_sigma(2.0)]), **cap_maaco_big(),
        **{"meta_" + k: v for k, v in META.items()})
    print("big cases", len(rows), "decodes", len(dec), "rebuilds", len(reb), f"{time.time() - t00:.0f} s")


if __name__ == "__main__":
    main()

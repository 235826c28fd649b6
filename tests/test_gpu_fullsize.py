"""BASELINE.json configs at (per-GPU) full size, checked through size-independent properties: every emitted
path is a legal start->target walk (8-connected, free cells, no corner cuts, self-avoiding where the
algorithm guarantees it), stored stats are reproduced by an independent numpy recomputation and by a
second scoring pass (idempotence), greedy steps are monotone, and the two A* variants agree on the optimal
length when no avoid set is involved."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SQ2 = math.sqrt(2.0)


def check_paths(grid, cells, lens, start, target, self_avoiding=False, sample=None):
    R, C = grid.shape
    occ = grid == 1
    idx = np.flatnonzero(lens > 0)
    if sample is not None and idx.size > sample:
        idx = idx[np.linspace(0, idx.size - 1, sample).astype(int)]
    out_len = {}
    for a in idx:
        p = cells[a, :lens[a]].astype(np.int64)
        r, c = p // C, p % C
        assert p[0] == start and p[-1] == target, a
        assert not occ[r, c].any(), a
        dr, dc = np.diff(r), np.diff(c)
        assert (np.maximum(np.abs(dr), np.abs(dc)) == 1).all(), a
        diag = (dr != 0) & (dc != 0)
        # corner-cut rule (helper.py:45-49): both orthogonal cells of a diagonal step are free
        assert not occ[r[1:][diag], c[:-1][diag]].any() and not occ[r[:-1][diag], c[1:][diag]].any(), a
        if self_avoiding:
            assert np.unique(p).size == p.size, a
        L = 0.0
        for d in diag:                                   # naive left-to-right sum, as the reference
            L = L + (SQ2 if d else 1.0)
        turns = int(((dr[1:] != dr[:-1]) | (dc[1:] != dc[:-1])).sum())
        out_len[int(a)] = (L, turns)
    return out_len


def test_cfg2_maaco_256_ants_128():
    import pathfit
    from pathfit import env
    g = env.bench_grid(128)
    m = pathfit.MAACO(g, 256, 100, 1.0, 7.0, 0.1, 2.5, 1.0, 0.9, 0.2, 0.9, 0.5, 0.1, seed=2)
    succ = []
    for it in range(1, 9):
        plen, turns = m.walk_iteration(it)
        dc, dl = m._bufs[1].download(), m._bufs[2].download()
        got = check_paths(g, dc, dl, 0, 128 * 128 - 1, self_avoiding=True, sample=64)
        for a, (L, T) in got.items():
            assert plen[a] == L and turns[a] == T
        assert ((dl == 0) == np.isinf(plen)).all()
        succ.append((dl > 0).mean())
        best = plen.min()
        m.best_path_length_overall = min(m.best_path_length_overall, best)
        m.update_pheromone()
        tau = m.pheromone_matrix
        bl = m.best_path_length_overall if np.isfinite(m.best_path_length_overall) else 256.0
        tmax = (1.0 / 0.9) * (1.0 / bl)
        assert (tau[g == 1] == 1e-9).all() and tau[g != 1].max() <= tmax and tau[g != 1].min() >= tmax / 256.0
    assert np.mean(succ) >= 0.5, succ                      # the cfg-2 grid must keep >= 50 % of the ants alive


@pytest.mark.parametrize("N,beta", [(4096, 2.0), (1024, 1.5)])
def test_cfg3_mpa_4096_predators_512(N, beta):
    import pathfit
    from pathfit import env
    from pathfit.engine import score_params
    g = env.bench_grid(512)
    # six iterations of a 6-iteration run: phases 1 (it 1-2), 2 (it 3-4: Levy on the prey / Brownian on the elite) and
    # 3 (it 5-6: Levy on the elite, CF -> 0) all execute at full size (MPA.py:339-377)
    m = pathfit.MPA(g, N, 6, FADs_rate=0.2, P_const=0.5, levy_beta=beta, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
                    min_safe_distance=1.8, diagonal_obstacle_penalty=100.0, seed=1)
    fit0 = m.d_stats.download()[:, 4].copy()
    for it in range(1, 7):
        m.step(it)
        cand_len, cand_cells, cand_stats = m.d_cand_len.download(), m.d_cand_cells.download(), m.d_cand_stats.download()
        got = check_paths(g, cand_cells, cand_len, 0, 512 * 512 - 1, sample=96)
        for a, (L, T) in got.items():
            assert cand_stats[a, 0] == L and cand_stats[a, 1] == T and cand_stats[a, 2] == 0.0
            assert cand_stats[a, 4] == L + 0.1 * T + 0.8 * 0.0 + cand_stats[a, 3]
        stats = m.d_stats.download()
        lens, cells = m.d_len.download(), m.d_cells.download()
        check_paths(g, cells, lens, 0, 512 * 512 - 1, sample=64)
        assert (stats[:, 4] <= fit0).all()                  # memory + FADs only ever improve (MPA.py:382,:402,:408)
        # idempotence: a second scoring pass over the stored population reproduces the stored stats
        e = m.engine
        d2 = e.buf((N, 5), np.float64)
        e._ck(e.L.pf_score_batch(e.h, m._sp, N, m.path_cap, m.d_cells.ptr, m.d_len.ptr, d2.ptr))
        assert np.array_equal(d2.download(), stats)
        assert (np.diff(stats[m.order, 4]) >= 0).all()      # population sorted by fitness (MPA.py:412)
        fit0 = stats[:, 4].copy()
    assert (m.d_status.download() != 3).all()


def test_cfg4_ga_pso_2048_per_gpu_512():
    import pathfit
    from pathfit import env
    from pathfit.engine import score_params
    g = env.bench_grid(512)
    e = pathfit.Engine(g)
    rnd = np.random.default_rng(4)
    free = np.flatnonzero(g.reshape(-1) != 1)
    n, W, cap = 2048, 5, 16 * 1024 + 64
    sp = score_params(0, True, 0.3, 0.8, 1.8, 100.0)
    wp = rnd.choice(free, (n, W)).astype(np.int32)
    paths, st, stats = e.decode_host(0, 512 * 512 - 1, wp_cells=wp, sp=sp, path_cap=cap)
    assert (st != 3).all() and (st == 0).sum() > n // 2
    lens = np.array([len(p) for p in paths]); cells = np.zeros((n, max(lens.max(), 1)), np.int32)
    for i, p in enumerate(paths):
        cells[i, :len(p)] = p
    got = check_paths(g, cells, lens, 0, 512 * 512 - 1, sample=64)
    for a, (L, T) in got.items():
        assert stats[a, 0] == L and stats[a, 1] == T
        assert all(w in set(paths[a].tolist()) for w in wp[a])        # every waypoint is visited, in order
    assert np.isinf(stats[lens == 0, 4]).all()
    # PSO: the same chromosomes as float positions (exact integers round to themselves)
    pos = np.stack([wp // 512, wp % 512], -1).astype(np.float64)
    paths2, st2, stats2 = e.decode_host(0, 512 * 512 - 1, wp_pos=pos, sp=sp, path_cap=cap)
    assert all(np.array_equal(a, b) for a, b in zip(paths, paths2)) and np.array_equal(stats, stats2)


def test_cfg4_solver_loops_2048_agents_512():
    """BASELINE.json configs[3] per-GPU share through the SOLVER LOOPS (not only the decode kernel): PSOSolver with 2048
    particles, two asynchronous sweeps, and one GASolver generation of 2048 individuals, on G512.  Properties: legal paths,
    stored stats == independent recomputation, pbest monotone, gbest == the first minimum of pbest, the speculate-and-repair
    rounds are invariant under the amount of speculation, the GA population is sorted and never loses its best."""
    import pathfit
    from pathfit import env
    g = env.bench_grid(512)
    S, T = 0, 512 * 512 - 1
    kw = dict(turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8, diagonal_obstacle_penalty_value=100.0)
    N, W = 2048, 5
    runs = []
    for spec in (None, 300):
        e = pathfit.Engine(g)
        ps = pathfit.PSOSolver(g, num_iterations=2, num_particles=N, num_waypoints_per_particle=W, w=0.7, c1=1.5, c2=1.5, engine=e,
                               seed=9, asynchronous=True, **kw)
        ps.max_speculation = spec
        assert ps.begin()
        pbf = [ps._d["pbf"].download().copy()]
        for _ in range(2):
            gf = ps.sweep()
            pbf.append(ps._d["pbf"].download().copy())
            assert (pbf[-1] <= pbf[-2]).all()                                   # pso.py:216 strict improvement only
            assert gf == pbf[-1].min() == ps.convergence_curve[-1]              # pso.py:222-229
        d = ps._d
        cells, lens, stats = d["cells"].download(), d["len"].download(), d["stats"].download()
        pbc, pbl = d["pb_cells"].download(), d["pb_len"].download()
        res = ps.finish()
        g_idx = int(np.argmin(pbf[-1]))                                         # first minimum
        assert res[5] == pbf[-1][g_idx]
        if spec is None:
            for a, (L, Tn) in check_paths(g, cells, lens, S, T, sample=48).items():
                assert stats[a, 0] == L and stats[a, 1] == Tn
            check_paths(g, pbc, pbl, S, T, sample=48)
            assert (lens > 0).sum() > N // 10                           # (a waypoint on an obstacle makes a particle infeasible, pso.py:77: ~20 % survive)
        runs.append((pbf[-1], d["pos"].download(), d["vel"].download(), d["pb"].download(), list(ps.convergence_curve), list(res[0]), res[5]))
        e.close()
    a, b = runs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert a[4] == b[4] and a[5] == b[5] and a[6] == b[6]

    e = pathfit.Engine(g)
    ga = pathfit.GASolver(g, num_generations=1, population_size=N, num_waypoints_per_chromosome=W, mutation_rate=0.1, crossover_rate=0.8,
                          engine=e, seed=5, **kw)
    res = ga.solve()
    pop = ga.population
    assert len(pop) == N
    fit = np.array([p["fitness"] for p in pop])
    assert (np.diff(fit) >= 0).all() and res[5] == fit[0] <= ga.convergence_curve[0]      # ga_solver.py:209-213
    C_ = 512
    for i in np.linspace(0, N - 1, 40).astype(int):
        p = pop[i]
        cp = np.array(p["path"].cells if hasattr(p["path"], "cells") else [r * C_ + c for r, c in p["path"]], np.int32)
        if np.isinf(p["fitness"]):
            assert cp.size == 0
            continue
        got = check_paths(g, cp[None, :], np.array([cp.size]), S, T)
        assert got[0] == (p["length"], p["turns"])
        at = 0                                              # ga_solver.py:58-93: the waypoints are visited in order (a waypoint may
        for r, c in p["chromosome"]:                        # also be crossed earlier: the segment's goal is exempt from the avoid set)
            hits = np.flatnonzero(cp[at:] == r * C_ + c)
            assert hits.size, (i, r, c)
            at += int(hits[0])
    e.close()


def test_cfg5_maaco_and_astar_8192_per_gpu_1024():
    import pathfit
    from pathfit import env
    g = env.bench_grid(1024)
    e = pathfit.Engine(g)
    m = pathfit.MAACO(g, 8192, 100, 1.0, 7.0, 0.1, 2.5, 1.0, 0.9, 0.2, 0.9, 0.5, 0.1, engine=e, seed=5)
    plen, turns = m.walk_iteration(1)
    dc, dl = m._bufs[1].download(), m._bufs[2].download()
    got = check_paths(g, dc, dl, 0, 1024 * 1024 - 1, self_avoiding=True, sample=48)
    for a, (L, T) in got.items():
        assert plen[a] == L and turns[a] == T
    assert (dl > 0).mean() > 0.3
    m.best_path_length_overall = plen.min()
    m.update_pheromone()
    # standalone connector batch (MAACO itself never calls A*): 8192 seeded pairs, both variants
    rnd = np.random.default_rng(6)
    free = np.flatnonzero(g.reshape(-1) != 1)
    n = 8192
    # keep pairs within 192 cells so the batch stays a few seconds
    s = rnd.choice(free, n)
    off = rnd.integers(-192, 193, (n, 2))
    tr = np.clip(s // 1024 + off[:, 0], 0, 1023); tc = np.clip(s % 1024 + off[:, 1], 0, 1023)
    t = (tr * 1024 + tc)
    t = np.where(g.reshape(-1)[t] == 1, s, t)
    p0, st0 = e.astar_host(0, s, t, path_cap=4096)
    p1, st1 = e.astar_host(1, s, t, path_cap=4096)
    assert (st0 != 3).all() and (st1 != 3).all() and ((st0 == 0) == (st1 == 0)).all()
    lens = np.array([len(p) for p in p0]); cells = np.zeros((n, max(lens.max(), 1)), np.int32)
    for i, p in enumerate(p0):
        cells[i, :len(p)] = p
    ok = np.flatnonzero(st0 == 0)[:64]
    for a in ok:
        r = check_paths(g, cells[a:a + 1], lens[a:a + 1], int(s[a]), int(t[a]))
    # both variants are optimal without an avoid set: equal path lengths (within fp summation order)
    def plen_of(p):
        d = np.diff(np.stack([p // 1024, p % 1024], 1).astype(np.int64), axis=0)
        return float(np.sqrt((d * d).sum(1).astype(np.float64)).sum())
    for a in np.flatnonzero(st0 == 0)[:512]:
        assert abs(plen_of(p0[a]) - plen_of(p1[a])) < 1e-6, a


def test_maaco_one_pass_update_equals_three_kernel_form_and_numpy_at_512():
    """maaco512 at its bench size: the iteration entry (in-walk marking into the stretch-blocked, flagged bit matrix, one-pass
    k_tau_update with its dense / sparse word forms) against (a) the three-kernel form the sharded fold uses, on a second
    handle, and (b) MAACO.py:304-332 restated in numpy over the downloaded paths: per cell the deposits of the visiting
    ants added one by one in ant order.  Three iterations, so marks, flags and tau carry over."""
    import pathfit
    from pathfit import env
    g = env.bench_grid(512)
    R = C = 512
    N = 16384
    kw = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
    m1 = pathfit.MAACO(g, N, 100, C0_initial_pheromone=0.1, seed=5, **kw)
    m2 = pathfit.MAACO(g, N, 100, C0_initial_pheromone=0.1, seed=5, **kw)
    tau_np = np.array(m1.pheromone_matrix, np.float64).reshape(-1)
    best = (float("inf"), float("inf"))
    for it in (1, 2, 3):
        m1.iterate_dev(it)
        # (a) the same iteration in separate steps on the other handle
        m2.walk_iteration_dev(it)
        dc, dl, dp, dt, ds = m2.walk_bufs()
        ib_len, ib_turns, ib_idx = m2.engine.maaco_best_dev(N, dp, dt)
        if ib_len < best[0]:
            best = (ib_len, ib_turns)
        elif abs(ib_len - best[0]) < 1e-9 and ib_turns < best[1]:
            best = (best[0], ib_turns)
        m2.engine.maaco_evaporate()
        m2.engine.maaco_deposit(N, m2.path_cap, dc, dl, dp)
        m2.engine.maaco_clip(best[0])
        t1, t2 = np.asarray(m1.pheromone_matrix), np.asarray(m2.pheromone_matrix)
        assert np.array_equal(t1, t2), it
        assert m1.best_path_length_overall == best[0]
        # (b) numpy, cell by cell in ant order (a cell is visited at most once per ant)
        cells, lens, plen = dc.download().reshape(N, -1), dl.download(), dp.download()
        tau_np = tau_np * (1.0 - kw["rho"])
        good = np.flatnonzero((lens > 0) & np.isfinite(plen) & (plen > 1e-6))
        ants = np.repeat(good, lens[good])
        flat = np.concatenate([cells[a, :lens[a]] for a in good]).astype(np.int64)
        order = np.lexsort((ants, flat))                                  # by cell, then by ant
        flat, dep = flat[order], (kw["Q"] / plen)[ants[order]]
        starts = np.flatnonzero(np.r_[True, flat[1:] != flat[:-1]])
        ends = np.r_[starts[1:], flat.size]
        rank = np.arange(flat.size) - np.repeat(starts, ends - starts)    # position of a deposit within its cell
        o2 = np.argsort(rank, kind="stable")
        flat, dep = flat[o2], dep[o2]
        cut = np.r_[0, np.cumsum(np.bincount(rank))]
        for k in range(cut.size - 1):                                     # round k: every cell's k-th deposit, one add each
            tau_np[flat[cut[k]:cut[k + 1]]] += dep[cut[k]:cut[k + 1]]
        bl = best[0] if np.isfinite(best[0]) else float(R + C)
        tmax = (1.0 / (1.0 - kw["rho"])) * (1.0 / max(bl, 1e-6))
        tmin = tmax / (2.0 * max(R, C))
        tau_np = np.where(g.reshape(-1) == 1, 1e-9, np.minimum(np.maximum(tau_np, tmin), tmax))
        assert np.array_equal(t1.reshape(-1), tau_np), it


MAACO_KW = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)


@pytest.mark.parametrize("G,N,seed,pack8_min,ants_per_wave,ahead", [(512, 16384, 0, None, 8, -1), (1024, 8192, 0, None, 8, -1), (1024, 8192, 3, 1, 8, 0),
                                                                    (512, 16384, 4, 1 << 30, 8, -1), (512, 5000, 6, 1, 1, -1), (512, 16384, 7, None, 8, 1)])
def test_maaco_full_bench_batches_vs_oracle(G, N, seed, pack8_min, ants_per_wave, ahead):
    """The bench batches themselves (maaco512: 16 384 ants on G512; maaco1024 = BASELINE configs[4]'s per-GPU share: 8 192 ants on
    G1024), iterations 1-3 through the one-enqueue iteration entry (pf_maaco_iterate), against the oracle restatement of
    MAACO.py:278-332 -- EVERY ant of every iteration, cell for cell: the oracle walks the same (seed, iteration, ant) streams over the
    pheromone matrix downloaded before the iteration, so a wrong-but-legal selection (a tie set, a roulette index, a q0 branch)
    anywhere in 600-1 700 steps shows.  The pheromone after each update must equal the oracle's sequential update over all paths.
    pack8_min None = the library's own choice of kernel for the batch; 1 / 2^30 force the packed / one-ant-per-wave kernels;
    ants_per_wave < 8 leaves groups of the packed kernel idle (5 000 ants, one group per wave, 4 096 waves: groups fetch a second ant);
    ahead -1 / 0 / 1 = the packed kernel's load-ahead form by occupancy (on for 8 192 ants, off for 16 384) / never / always."""
    import pathfit
    from pathfit import env
    import pf_oracle as po
    g = env.bench_grid(G)
    o = po.Oracle(g)
    s, t = 0, G * G - 1
    m = pathfit.MAACO(g, N, 100, C0_initial_pheromone=0.1, seed=seed, **MAACO_KW)
    if pack8_min is not None:
        m.engine.set_option("maaco_pack8_min", pack8_min)
    m.engine.set_option("maaco_load_ahead", ahead)
    m.engine.set_option("maaco_ants_per_wave", ants_per_wave)     # (1: groups 1..7 of every wavefront stay idle; 5 000 ants on 4 096 waves: groups refetch inside the loop)
    try:
        P = po.MaacoParams(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn=1.0, wh_max=0.9, wh_min=0.2, k_h=0.9, q0_initial=0.5, C0=0.1,
                           num_iterations=100)
        tau0, dist = o.maaco_init(s, t, 0.1)
        assert np.array_equal(tau0.reshape(G, G), m.pheromone_matrix)
        best = float("inf")
        dead = longest = 0
        for it in (1, 2, 3):
            tau = np.ascontiguousarray(np.asarray(m.pheromone_matrix, np.float64).reshape(-1))
            m.iterate_dev(it)
            dc, dl, dp, dt, ds = m.walk_bufs()
            cells, lens, plen, turns, st = dc.download().reshape(N, -1), dl.download(), dp.download(), dt.download(), ds.download()
            paths, olens = [], []
            for ant in range(N):
                p, L, T, _ = o.maaco_walk(s, t, P, tau, dist, it, seed, ant)
                if len(p) == 0:
                    assert lens[ant] == 0 and st[ant] in (1, 2) and np.isinf(plen[ant]) and turns[ant] == -1, (it, ant, st[ant])
                    dead += 1
                else:
                    assert lens[ant] == len(p) and np.array_equal(cells[ant, :len(p)], p), (it, ant)
                    assert plen[ant] == L and turns[ant] == T and st[ant] == 0, (it, ant)
                    longest = max(longest, len(p))
                paths.append(p); olens.append(L)
            best = min(best, min(olens))
            o.maaco_update(tau, 0.1, 2.5, paths, olens, best)
            assert m.best_path_length_overall == best
            assert np.array_equal(np.asarray(m.pheromone_matrix).reshape(-1), tau), it
        assert dead > N // 10 and longest > 1.5 * G        # ants that die in dead ends and walks far longer than the diagonal both occur
    finally:
        m.engine.set_option("maaco_pack8_min", 2048)
        m.engine.set_option("maaco_ants_per_wave", 8)
        m.engine.set_option("maaco_load_ahead", -1)


def test_cfg3_mpa_sweeps_4096_512_vs_oracle():
    """BASELINE configs[2] at full size against the oracle: a 6-iteration run of 4 096 predators on G512 (phases 1, 2 and 3 of
    MPA.py:339-377 execute, main.py:44-52 parameters).  Before every iteration the sorted population is downloaded and handed to
    the oracle's restatement of the loop body; for a sample of 112 predators per iteration -- strided over the sorted list plus the
    first and the last sixteen -- the candidate of the phase sweep (path, five stats; wherever the device did not prove it
    rejected), and the individual after memory + FADs (path, five stats) must be the oracle's, bit for bit."""
    import pathfit
    from pathfit import env
    import pf_oracle as po
    import pf_loops
    g = env.bench_grid(512)
    N, K, seed = 4096, 6, 1
    S, T = 0, 512 * 512 - 1
    m = pathfit.MPA(g, N, K, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
                    min_safe_distance=1.8, diagonal_obstacle_penalty=100.0, seed=seed)
    o = po.Oracle(g)
    ref = pf_loops.MpaOracle(o, S, T, N, K, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1, w_safe=0.8, min_safe=1.8,
                             diag_pen=100.0, seed=seed)
    sample = sorted(set(list(range(16)) + list(range(N - 16, N)) + list(range(5, N, 51))))
    rebuilt = changed = 0
    for it in range(1, K + 1):
        m._sort()                                             # (idempotent: step() starts with the same stable sort, MPA.py:333)
        order = m.order.copy()
        cells, lens, stats = m.d_cells.download(), m.d_len.download(), m.d_stats.download()
        ref.pop = [(cells[sl, :lens[sl]].copy(), stats[sl].copy()) for sl in order]
        elite = ref.pop[0]
        ratio = it / K
        CF = 0.0 if ratio >= 1.0 else (1.0 - ratio) ** (2.0 * ratio)
        m.step(it)
        c_len, c_cells, c_stats, c_st = m.d_cand_len.download(), m.d_cand_cells.download(), m.d_cand_stats.download(), m.d_status.download()
        cells2, lens2, stats2 = m.d_cells.download(), m.d_len.download(), m.d_stats.download()
        for i in sample:
            cand = ref.phase_candidate(it, i, elite, CF)
            if c_st[i] == 0:                                  # rebuilt on the device: the candidate itself is compared
                assert np.array_equal(c_cells[i, :c_len[i]], cand[0]) and np.array_equal(c_stats[i], cand[1]), (it, i)
                rebuilt += 1
            else:                                             # unmodified, failed or proven rejected: the oracle's candidate must not be accepted either
                assert c_st[i] == 4 and not (cand[1][4] < ref.pop[i][1][4]) or np.array_equal(c_cells[i, :c_len[i]], cand[0]), (it, i, c_st[i])
            ind = cand if cand[1][4] < ref.pop[i][1][4] else ref.pop[i]          # memory, MPA.py:381-384
            ind = ref.fads(it, i, ind, CF)                                        # FADs, :387-410
            sl = order[i]
            assert np.array_equal(cells2[sl, :lens2[sl]], ind[0]) and np.array_equal(stats2[sl], ind[1]), (it, i)
            changed += not np.array_equal(ind[0], ref.pop[i][0])
    assert rebuilt > 100 and changed > 20, (rebuilt, changed)

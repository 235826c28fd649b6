"""End-to-end: the GPU solver facades == the same outer loops driven by the CPU oracle
(oracle/pf_loops.py), bit for bit, including BASELINE.json config-1
(MPA, 30 predators, 50 iterations, fig7 20x20)."""
import ctypes as C
import math

import numpy as np
import pytest

import golden_io as gio

pytestmark = pytest.mark.gpu


def test_cfg1_mpa_30x50_fig7_matches_oracle_loop():
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    for seed in (0, 1, 2):
        m = pathfit.MPA(g, 30, 50, seed=seed)
        got = m.solve_path_planning()
        ref = pf_loops.MpaOracle(po.Oracle(g), s, t, 30, 50, seed=seed)
        best = ref.solve()
        assert [r * 20 + c for r, c in got[0]] == list(best[0]), seed
        assert (got[1], got[2], got[3], got[4], got[5]) == (best[1][0], int(best[1][1]), best[1][2], best[1][3], best[1][4])
        assert m.convergence_curve_data == ref.curve
        # whole population too
        pop = m.population
        for a, b in zip(pop, ref.pop):
            assert np.array_equal(a["path"].cells, b[0]) and a["fitness"] == b[1][4]
        # reference's measured range for this config: 32.86 - 33.04
        assert 31.0 < got[5] < 34.5


def test_mpa_main_params_with_levy_phases():
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("img1")
    kw = dict(FADs_rate=0.2, P_const=0.5, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
              min_safe_distance=1.8, diagonal_obstacle_penalty=100.0)
    m = pathfit.MPA(g, 24, 12, seed=5, fused=False, **kw)      # the three separate ABI calls
    got = m.solve_path_planning()
    ref = pf_loops.MpaOracle(po.Oracle(g), s, t, 24, 12, FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1,
                             w_safe=0.8, min_safe=1.8, diag_pen=100.0, seed=5)
    best = ref.solve()
    assert [r * 20 + c for r, c in got[0]] == list(best[0]) and got[5] == best[1][4]
    assert m.convergence_curve_data == ref.curve


@pytest.mark.parametrize("fused", [True, False])
def test_mpa_bound_pruning_changes_nothing(fused):
    """The exact length-bound pruning of rebuilds must leave every predator, every iteration, every
    phase exactly as without it -- and must actually fire on a 256 x 256 map."""
    import pathfit
    g, s, t = gio.grid("g256")
    kw = dict(FADs_rate=0.2, P_const=0.5, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
              min_safe_distance=1.8, diagonal_obstacle_penalty=100.0)
    runs = []
    for prune in (1, 0):
        m = pathfit.MPA(g, 96, 9, seed=11, fused=fused, **kw)      # 3 iterations in each phase
        m.engine.set_option("mpa_prune", prune)
        pruned = 0
        for it in range(1, 10):
            m.step(it)
            pruned += m.engine.counters()["pruned_rebuilds"]
        pop = m.population
        runs.append(([list(p["path"].cells) for p in pop], [p["fitness"] for p in pop], list(m.convergence_curve_data), pruned))
        m.engine.set_option("mpa_prune", 1)
        m.engine.close()
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] == runs[1][2]
    assert runs[0][3] > 0 and runs[1][3] == 0


@pytest.mark.parametrize("gname,N", [("g256", 384), ("g512", 1024)])
def test_two_wave_searches_equal_single_wave(gname, N):
    """pf_astar_pr.h: a search on a pop wave + a pool wave pops exactly what the single wave pops -- whole populations,
    candidate rows, pop / push counters of every sweep, three phases."""
    import pathfit
    from pathfit import env
    g = env.bench_grid(512) if gname == "g512" else gio.grid("g256")[0]
    kw = dict(FADs_rate=0.2, P_const=0.5, levy_beta=1.5, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
              min_safe_distance=1.8, diagonal_obstacle_penalty=100.0)
    runs = []
    for two in (1, 0):
        m = pathfit.MPA(g, N, 6, seed=3, **kw)
        try:
            m.engine.set_option("two_wave", two)
        except pathfit.PathfitError as ex:                    # the default build leaves the engine out (it measured 0.90x): PF_EXTRA_FLAGS=-DPF_TWO_WAVE
            assert "not built in" in str(ex)
            pytest.skip("libpathfit.so was built without -DPF_TWO_WAVE")
        log = []
        for it in range(1, 7):
            m.step(it)
            c = m.engine.counters()
            log.append((c["pops"], c["pushes"], c["nbr_examined"], c["overflow_agents"]))
        runs.append((m.d_cells.download(), m.d_len.download(), m.d_stats.download(), m.d_cand_len.download(), m.d_cand_stats.download(), log))
        m.engine.set_option("two_wave", 0)
        m.engine.close()
    a, b = runs
    assert a[5] == b[5], (a[5], b[5])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
    for i in range(N):
        assert np.array_equal(a[0][i, :a[1][i]], b[0][i, :b[1][i]])
    assert all(x[3] == 0 for x in a[5])


@pytest.mark.parametrize("beta", [7.0, 2.0])
def test_maaco_solve_matches_oracle_loop(beta):
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    kw = dict(alpha=1.0, beta=beta, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9,
              q0_initial=0.5)
    m = pathfit.MAACO(g, 50, 12, C0_initial_pheromone=0.1, seed=3, **kw)
    path, length, turns = m.solve_path_planning()
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, 50, 12, C0=0.1, seed=3, **kw)
    assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"]
    assert m.convergence_curve_data == ref["curve"]
    assert np.array_equal(m.pheromone_matrix, ref["tau"])


@pytest.mark.parametrize("Q", [2.5, 110.0, 4000.0, 1e-12])
def test_maaco_dense_deposit_paths_any_deposit_size(Q):
    """The pheromone pass adds the deposits of a cell's ants in ant order; where more than a dozen of a 64-ant word visit a cell it
    steps through all 64 with the ant's bit turned into a double -- by default as 0.0 / 2^(2^k - 1023) against deposits pre-scaled by
    the inverse power of two (exact for deposits below 4), else as 0.0 / 1.0.  400 ants on the 20 x 20 map make every word around the
    start dense; Q = 2.5 takes the scaled form, Q = 110 puts deposits on both sides of the limit (the shortest possible walk is 26.9
    long: Q / L reaches 4.09; a block whose chunk holds one deposit of 4 or more falls back as a whole), Q = 4000 takes the 0.0 / 1.0
    form, Q = 1e-12 scales tiny deposits up by 2^1022.  Pheromone, path and curve must equal the oracle loop bit for bit."""
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    kw = dict(alpha=1.0, beta=7.0, rho=0.1, Q=Q, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
    m = pathfit.MAACO(g, 400, 3, C0_initial_pheromone=0.1, seed=11, **kw)
    path, length, turns = m.solve_path_planning()
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, 400, 3, C0=0.1, seed=11, **kw)
    assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"]
    assert m.convergence_curve_data == ref["curve"]
    assert np.array_equal(m.pheromone_matrix, ref["tau"])


@pytest.mark.parametrize("n_ants", [50, 3000])
def test_maaco_path_rows_too_short_redo_leaves_pheromone_untouched(n_ants):
    """The overflow-redo branch of the one-enqueue iteration (pf_maaco_iterate): with path rows of 8 cells every iteration's first
    attempt overflows, k_tau_update must return without moving tau (skipped = 1; the bit matrix and its flags are left dirty and
    are wiped by the redo), and the facade repeats the iteration with longer rows.  Result, curve and pheromone must equal the
    oracle loop bit for bit, in both walk kernels (50 ants: one per wave; 3000: eight per wave), and path_cap must have grown.
    Then the order walk(A), deposit(B), deposit(A): the second deposit of A's paths must not trust marks a previous pass consumed."""
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    kw = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
    m = pathfit.MAACO(g, n_ants, 4, C0_initial_pheromone=0.1, seed=3, **kw)
    m.path_cap = 8
    path, length, turns = m.solve_path_planning()
    assert m.path_cap > 8
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, n_ants, 4, C0=0.1, seed=3, **kw)
    assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"]
    assert m.convergence_curve_data == ref["curve"]
    assert np.array_equal(m.pheromone_matrix, ref["tau"])
    # walk(A) marks A's deposits; deposit(B) consumes / replaces the marks; deposit(A) must mark again (not deposit nothing)
    e = m.engine
    m.walk_iteration_dev(5)
    dc, dl, dp, dt, ds = m.walk_bufs()
    cells, lens, plen = dc.download(), dl.download(), dp.download()
    db_c, db_l, db_p = e.put(cells[: n_ants // 2].copy()), e.put(lens[: n_ants // 2].copy()), e.put(plen[: n_ants // 2].copy())
    tau0 = np.array(m.pheromone_matrix, np.float64)
    e.maaco_deposit(n_ants // 2, m.path_cap, db_c, db_l, db_p)
    tau1 = np.array(m.pheromone_matrix, np.float64)
    e.maaco_deposit(n_ants, m.path_cap, dc, dl, dp)
    tau2 = np.array(m.pheromone_matrix, np.float64).reshape(-1)
    o = po.Oracle(g)
    want = tau0.reshape(-1).copy()
    for paths, ls in (([cells[a, :lens[a]] for a in range(n_ants // 2)], plen[: n_ants // 2]), ([cells[a, :lens[a]] for a in range(n_ants)], plen)):
        for p, L in zip(paths, ls):                           # MAACO.py:306-311 alone (no evaporation, no clip): sequential adds
            if len(p) and np.isfinite(L) and L > 1e-6:
                want[p] += 2.5 / L
    assert not np.array_equal(tau1, tau0) and np.array_equal(tau2, want)


def test_maaco_20000_ants_deposit_in_two_chunks():
    """More than 16 384 ants: the ordered deposit stages the ants' values in LDS one 16 384-ant chunk at a time; the
    pheromone matrix must still be the sequential sum of MAACO.py:306-311 bit for bit."""
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    kw = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9,
              q0_initial=0.5)
    m = pathfit.MAACO(g, 20000, 2, C0_initial_pheromone=0.1, seed=3, **kw)
    path, length, turns = m.solve_path_planning()
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, 20000, 2, C0=0.1, seed=3, **kw)
    assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"]
    assert np.array_equal(m.pheromone_matrix, ref["tau"])


def test_maaco_alpha_not_one_uses_host_pow_table():
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig13")
    kw = dict(alpha=1.5, beta=3.0, rho=0.2, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.3)
    m = pathfit.MAACO(g, 24, 5, seed=9, **kw)
    path, length, turns = m.solve_path_planning()
    ref = pf_loops.maaco_solve(po.Oracle(g), s, t, 24, 5, seed=9, **kw)
    assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"]
    assert np.array_equal(m.pheromone_matrix, ref["tau"])


def _oracle_backed(cls, orc):
    """Same facade, but decode+score from the CPU oracle (checker only)."""
    from pathfit.paths import CellPath

    class OB(cls):
        def _evaluate(self, wp_cells=None, wp_pos=None):
            n = len(wp_cells) if wp_cells is not None else len(wp_pos)
            cps, stats, feas = [], np.zeros((n, 5)), np.zeros(n, bool)
            for i in range(n):
                wp = wp_cells[i] if wp_cells is not None else orc.pso_round(wp_pos[i])
                p, _ = orc.decode(self._cell(self.start_node), self._cell(self.target_node), wp)
                sp = self._sp
                stats[i] = orc.score(p, 0, sp.w_turn, sp.w_safe, sp.min_safe, bool(sp.restrict_policy), sp.diag_pen)
                cps.append(CellPath(p, self.cols)); feas[i] = len(p) > 0
            return cps, stats, feas
    return OB


def test_ga_solve_matches_oracle_backed_facade():
    import pathfit, pf_oracle as po
    g, s, t = gio.grid("fig7")
    kw = dict(num_generations=6, population_size=24, num_waypoints_per_chromosome=5, mutation_rate=0.1, crossover_rate=0.8,
              tournament_size=3, turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
              diagonal_obstacle_penalty_value=100.0, seed=4)
    a = pathfit.GASolver(g, **kw)
    ra = a.solve()
    b = _oracle_backed(pathfit.GASolver, po.Oracle(g))(g, engine=a.engine, **kw)
    rb = b.solve()
    assert ra == rb and a.convergence_curve == b.convergence_curve
    assert ra[0][0] == (0, 0) and ra[0][-1] == (19, 19)


def test_pso_solve_matches_oracle_loop():
    import pathfit, pf_oracle as po
    g, s, t = gio.grid("fig7")
    orc = po.Oracle(g)
    kw = dict(num_iterations=8, num_particles=32, num_waypoints_per_particle=5, w=0.7, c1=1.5, c2=1.5,
              turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8, diagonal_obstacle_penalty_value=100.0,
              seed=6, asynchronous=False)
    a = pathfit.PSOSolver(g, **kw)
    ra = a.solve()
    # oracle loop: same init (host RNG), then synchronous sweeps
    b = _oracle_backed(pathfit.PSOSolver, orc)(g, engine=a.engine, **kw)
    assert b._initialize_particles()
    pos, vel, pb, pbf = b._pos.copy(), b._vel.copy(), b._pbest.copy(), b._pbest_fit.copy()
    gb, gfit, gpath = np.array(b.gbest_particle_data["position"]), b.gbest_particle_data["fitness"], b.gbest_particle_data["path"]
    curve = [gfit]
    for it in range(8):
        pos, vel = orc.pso_update(pos, vel, pb, gb, 0.7, 1.5, 1.5, b.max_vel, 6, it, 0)
        cps, stats, feas = b._evaluate(wp_pos=pos)
        imp = feas & (stats[:, 4] < pbf)
        pb[imp] = pos[imp]; pbf[imp] = stats[imp, 4]
        cand = np.flatnonzero(imp)
        if cand.size:
            j = cand[np.argmin(stats[cand, 4])]
            if stats[j, 4] < gfit:
                gfit, gb, gpath = stats[j, 4], pos[j].copy(), cps[j]
        curve.append(gfit)
    assert a.convergence_curve == curve and ra[5] == gfit
    assert ra[0] == (gpath.tolist() if hasattr(gpath, "tolist") else gpath)
    assert np.array_equal(a._pos, pos) and np.array_equal(a._pbest_fit, pbf)


def test_astar_solver_single_query():
    import pathfit, pf_oracle as po
    g, s, t = gio.grid("fig7")
    orc = po.Oracle(g)
    a = pathfit.AStarSolver(g, 0.3, 0.8, 1.8, True, True, 100.0)
    res = a.solve()
    want, _ = orc.astar(s, t, None, 0)
    assert [r * 20 + c for r, c in res[0]] == list(want)
    assert list(res[1:]) == [v if i != 1 else int(v) for i, v in enumerate(orc.score(want, 0, 0.3, 0.8, 1.8, True, 100.0))]
    assert a.solve((0, 4), (3, 3))[0] == [] and a.solve((0, 4), (3, 3))[5] == math.inf      # obstacle start -> []
    assert a.solve((2, 2), (2, 2))[0] == [(2, 2)]


def _sharded_mpa_worker(rank, world, port, out_dir):
    import os, sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np, pathfit, golden_io as gio
    from pathfit.dist import Comm, ShardedMPA
    g, s, t = gio.grid("fig7")
    eng = pathfit.Engine(g, device=0)                      # both ranks share GPU 0: exchange logic under test
    sm = ShardedMPA(Comm(dist, None), lambda n: pathfit.MPA(g, 30, 12, engine=eng, seed=3, n_local=n), 30)
    fits = [sm.step(it) for it in range(1, 13)]
    m = sm.local
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), fits=np.array(fits), stats=m.d_stats.download(), lens=m.d_len.download(),
             cells=m.d_cells.download(), gorder=sm.gorder)
    dist.barrier(); dist.destroy_process_group()


def test_sharded_mpa_two_ranks_equals_single(tmp_path):
    """1-GPU == N-GPU: two gloo ranks (each owning 15 of 30 predators, engines on the same GPU) reproduce the
    single-process population bit for bit after 12 iterations (all three MPA phases)."""
    import os
    import torch.multiprocessing as mp
    import pathfit
    from pathfit.dist import Comm, ShardedMPA
    g, s, t = gio.grid("fig7")
    sm = ShardedMPA(Comm(None), lambda n: pathfit.MPA(g, 30, 12, seed=3, n_local=n), 30)
    fits = [sm.step(it) for it in range(1, 13)]
    ref_stats, ref_len, ref_cells = sm.local.d_stats.download(), sm.local.d_len.download(), sm.local.d_cells.download()
    port = 29800 + os.getpid() % 1000
    mp.spawn(_sharded_mpa_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    assert np.array_equal(z0["fits"], fits) and np.array_equal(z1["fits"], fits)
    assert np.array_equal(z0["gorder"], sm.gorder) and np.array_equal(z1["gorder"], sm.gorder)
    stats = np.concatenate([z0["stats"], z1["stats"]]); lens = np.concatenate([z0["lens"], z1["lens"]])
    cells = np.concatenate([z0["cells"], z1["cells"]])
    assert np.array_equal(stats, ref_stats) and np.array_equal(lens, ref_len)
    for i in range(30):
        assert np.array_equal(cells[i, :lens[i]], ref_cells[i, :ref_len[i]])
    # and the single-rank sharded loop equals the plain facade
    m = pathfit.MPA(g, 30, 12, seed=3)
    for it in range(1, 13):
        m.step(it)
    assert np.array_equal(m.d_stats.download()[m.order], ref_stats[sm.gorder])


def test_maaco_engine_reused_with_other_ant_counts_and_abandoned_marks():
    """One engine, several colonies in a row with different ant counts (the deposit bit matrix and its chunk flags keep the
    capacity of the largest; a smaller colony uses a prefix of every stretch's words), one of them abandoned after a walk
    that marked its deposits but never updated: each colony must still equal the oracle's loop bit for bit."""
    import pathfit, pf_oracle as po, pf_loops
    g, s, t = gio.grid("fig7")
    kw = dict(alpha=1.0, beta=7.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
    eng = pathfit.Engine(g)
    for n_ants, iters, abandon in ((200, 3, False), (70, 4, True), (130, 3, False), (64, 2, False), (1, 3, False), (333, 2, False)):
        m = pathfit.MAACO(g, n_ants, iters, C0_initial_pheromone=0.1, seed=11, engine=eng, **kw)
        path, length, turns = m.solve_path_planning()
        ref = pf_loops.maaco_solve(po.Oracle(g), s, t, n_ants, iters, C0=0.1, seed=11, **kw)
        assert [r * 20 + c for r, c in path] == list(ref["path"]) and length == ref["length"] and turns == ref["turns"], n_ants
        assert np.array_equal(m.pheromone_matrix, ref["tau"]), n_ants
        if abandon:
            m.walk_iteration_dev(iters + 1)          # marks made, no update: the next colony must not see them

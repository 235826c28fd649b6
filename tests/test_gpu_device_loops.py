"""Iteration control in HBM and the multi-GPU exchange (8e), on the real engine:
device sorts / scans / GA operators against their host definitions, device-to-host copy budgets of the solver loops,
RCCL bound directly (single-rank communicator on the one GPU of the box), and 2-rank runs of all four sharded solvers
(two gloo processes sharing GPU 0: the exchange logic is transport independent) == the single-process result."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import golden_io as gio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MK = dict(alpha=1.0, beta=2.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
GA_KW = dict(num_generations=5, population_size=24, num_waypoints_per_chromosome=5, mutation_rate=0.1, crossover_rate=0.8,
             tournament_size=3, turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
             diagonal_obstacle_penalty_value=100.0)
PSO_KW = dict(num_iterations=6, num_particles=24, num_waypoints_per_particle=5, w=0.7, c1=1.5, c2=1.5, turn_penalty_factor=0.3,
              safety_penalty_factor=0.8, min_safe_distance=1.8, diagonal_obstacle_penalty_value=100.0)


def _engine(name="fig7"):
    import pathfit
    g, s, t = gio.grid(name)
    return pathfit.Engine(g), g, s, t


def test_device_sort_scan_view_helpers():
    from pathfit.dist import maaco_best_scan_host
    e, g, s, t = _engine()
    rnd = np.random.default_rng(0)
    # list.sort(key=fitness): stable, from the CURRENT list order; plenty of exact ties and +inf
    # (the hand-written rank sort: k_sort_prep / k_rank_count / k_rank_scatter.  Sizes around the 64-element tiles and the 8-key
    # scalar loads, beyond one 16 384-element LDS-sized block, few distinct keys -- long runs of equal keys whose order must be the
    # CURRENT list order --, all keys equal, negative values, -0.0 == 0.0, +inf last)
    for n in (1, 2, 7, 8, 9, 33, 63, 64, 65, 127, 129, 1000, 4096, 8191, 16384, 20000):
        for kind in ("ties", "two", "same", "signed", "distinct"):
            vals = rnd.random((n, 5))
            if kind == "ties":
                vals[:, 4] = np.round(rnd.random(n) * 8) / 4.0
                vals[rnd.integers(0, n, max(1, n // 10)), 4] = np.inf
            elif kind == "two":
                vals[:, 4] = rnd.integers(0, 2, n) * 3.5
            elif kind == "same":
                vals[:, 4] = 815.0458
            elif kind == "signed":
                vals[:, 4] = np.round(rnd.standard_normal(n) * 2) / 2.0
                vals[rnd.integers(0, n, max(1, n // 8)), 4] = -0.0
                vals[rnd.integers(0, n, max(1, n // 8)), 4] = 0.0
                vals[rnd.integers(0, n, max(1, n // 16)), 4] = -np.inf
            order = rnd.permutation(n).astype(np.int32)
            d_v, d_o = e.put(vals), e.put(order)
            e.sort_order_by_key(n, d_v, 5, 4, d_o)
            want = order[np.argsort(vals[order, 4], kind="stable")]
            assert np.array_equal(d_o.download(), want), (n, kind)
    # MAACO.py:343-349 on the device == the sequential scan (near-ties within 1e-9, failed ants, nobody arrives)
    for trial in range(60):
        n = int(rnd.integers(1, 3000))
        plen = np.round(rnd.random(n) * 20) / 2.0 + 50.0
        plen += rnd.choice([0.0, 3e-10, -4e-10, 8e-10], n)
        turns = rnd.integers(0, 6, n).astype(np.int32)
        dead = rnd.random(n) < (1.0 if trial == 0 else 0.2)
        plen[dead] = np.inf; turns[dead] = -1
        got = e.maaco_best_dev(n, e.put(plen), e.put(turns))
        want = maaco_best_scan_host(plen, turns)
        assert got == (want[0], want[1], want[2]), (trial, got, want)
    # positions / slots of the ids a rank stores
    N = 1000
    gorder = rnd.permutation(N).astype(np.int32)
    lo, hi = 337, 702
    d_g, d_i, d_s = e.put(gorder), e.buf(hi - lo, np.int32), e.buf(hi - lo, np.int32)
    e.mpa_local_view(N, d_g, lo, hi, d_i, d_s)
    pos = np.flatnonzero((gorder >= lo) & (gorder < hi))
    assert np.array_equal(d_i.download(), pos) and np.array_equal(d_s.download(), gorder[pos] - lo)
    # pbest -> gbest scan
    for trial in range(40):
        n = int(rnd.integers(1, 700))
        stats = rnd.random((n, 5)); stats[:, 4] = np.round(rnd.random(n) * 10) / 2.0
        lens = (rnd.random(n) < 0.8).astype(np.int32) * 7
        st = (rnd.random(n) < 0.02).astype(np.int32) * 3
        pbf = np.round(rnd.random(n) * 10) / 2.0
        gfit = 2.0
        imp = (lens > 0) & (stats[:, 4] < pbf) & (stats[:, 4] < gfit)
        d = [e.put(stats), e.put(lens), e.put(st), e.put(pbf)]
        for sync in (0, 1):
            idx, fit, ovf = e.pso_scan(n, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, gfit, sync)
            cand = np.flatnonzero(imp)
            want = -1 if not cand.size else (int(cand[0]) if not sync else int(cand[np.argmin(stats[cand, 4])]))
            assert idx == want and ovf == int((st == 3).sum()), (trial, sync)
            assert fit == (np.inf if want < 0 else stats[want, 4])


def test_ga_device_operators_match_host_native():
    """k_ga_select / k_ga_breed == pf_ga_select / pf_ga_breed (which tests/test_ga_native.py pins against the Python
    operators of ga_solver.py:136-160)."""
    from pathfit import solvers
    e, g, s, t = _engine("img2")
    rnd = np.random.default_rng(5)
    free = np.flatnonzero(g.reshape(-1) != 1)
    for N, W, k, cx, mut in ((6, 1, 3, 0.9, 0.5), (21, 2, 3, 0.8, 0.2), (22, 5, 3, 0.8, 0.2), (64, 5, 6, 0.5, 0.9), (63, 3, 7, 1.0, 0.0),
                             (40, 4, 1, 0.0, 1.0), (5, 5, 9, 0.7, 0.3), (257, 5, 3, 0.8, 0.1)):
        for seed in (0, 99):
            fit_sorted = np.sort(np.round(rnd.random(N) * 6) / 2.0)
            gorder = rnd.permutation(N).astype(np.int32)                     # list position -> storage id
            fit_all = np.zeros(N); fit_all[gorder] = fit_sorted
            chrom_all = rnd.choice(free, (N, W)).astype(np.int32)
            for gen in (0, 3):
                pidx = solvers.ga_select_native(seed, gen, fit_sorted, k)     # list positions
                d_psid = e.buf(N, np.int32)
                e.ga_select(seed, gen, N, k, e.put(fit_all), e.put(gorder), d_psid)
                assert np.array_equal(d_psid.download(), gorder[pidx]), (N, W, k, seed, gen)
                kids = solvers.ga_breed_native(seed, gen, cx, mut, g == 1, chrom_all[gorder[pidx]])
                for lo, hi in ((0, N), (1, N), (N // 3, 2 * N // 3 + 1)):
                    out = e.buf((max(hi - lo, 1), W), np.int32)
                    e.ga_breed(seed, gen, N, W, cx, mut, e.put(chrom_all), d_psid, lo, hi - lo, out)
                    assert np.array_equal(out.download()[: hi - lo], kids[lo:hi]), (N, W, k, seed, gen, lo, hi)


def test_solver_loops_keep_everything_in_hbm():
    """an MPA iteration, a GA generation and a PSO sweep make only small (<= 128 B) device-to-host
    copies -- counters, list heads, scan results -- and no bulk copy at all."""
    import pathfit
    g, s, t = gio.grid("fig7")
    m = pathfit.MPA(g, 64, 9, seed=2)
    e = m.engine
    a0 = e.d2h_counts()
    for it in range(1, 10):
        m.step(it)
    a1 = e.d2h_counts()
    assert a1[1] == a0[1], ("MPA bulk copies", a0, a1)
    assert a1[0] - a0[0] <= 9 * 8
    ga = pathfit.GASolver(g, seed=4, engine=e, **GA_KW)
    assert ga._initialize_population()
    ga.best_solution_overall = ga.population[0].copy()
    b0 = e.d2h_counts()
    ga.num_generations = 4
    ga._solve_device()
    b1 = e.d2h_counts()
    # the only bulk copies allowed are the path row of an improved best (<= one per generation)
    assert b1[1] - b0[1] <= 4 and b1[2] - b0[2] <= 4 * ga._gd["cap"] * 4, ("GA bulk", b0, b1)
    ps = pathfit.PSOSolver(g, seed=6, engine=e, **PSO_KW)
    assert ps.begin()
    c0 = e.d2h_counts()
    for _ in range(4):
        ps.sweep()
    c1 = e.d2h_counts()
    assert c1[1] == c0[1], ("PSO bulk copies", c0, c1)


def test_rccl_bound_directly_single_rank_communicator():
    """pf_comm_* really are RCCL: a one-rank communicator on the box's GPU runs every collective the solvers use
    (multi-rank RCCL needs one GPU per rank: the driver's multi-GPU bench is the first place that can run it)."""
    e, g, s, t = _engine()
    idb = (C.c_char * 128)()
    assert e.L.pf_comm_unique_id(idb) == 0, e.L.pf_last_error(None)
    e._ck(e.L.pf_comm_init(e.h, 0, 1, idb))
    assert e.L.pf_comm_world(e.h) == 1 and e.L.pf_comm_rank(e.h) == 0
    a = np.arange(1000, dtype=np.float64)
    src, dst = e.put(a), e.buf(1000, np.float64)
    e._ck(e.L.pf_comm_all_gather(e.h, src.ptr, dst.ptr, a.nbytes))
    e._ck(e.L.pf_comm_broadcast(e.h, dst.ptr, a.nbytes, 0))
    e._ck(e.L.pf_comm_all_reduce_f64(e.h, dst.ptr, 1000, 0))
    e._ck(e.L.pf_comm_all_reduce_f64(e.h, dst.ptr, 1000, 2))
    assert np.array_equal(dst.download(), a)
    dst2 = e.buf(1000, np.float64)
    e._ck(e.L.pf_comm_sendrecv(e.h, src.ptr, a.nbytes, 0, dst2.ptr, a.nbytes, 0))      # a ring step onto itself
    assert np.array_equal(dst2.download(), a)
    e._ck(e.L.pf_comm_destroy(e.h))


# ---- two gloo ranks sharing GPU 0 ------------------------------------------------------------------------------------
def _worker(rank, world, port, out_dir, what):
    for p in (os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pathfit
    from pathfit.dist import Comm, ShardedMAACO, ShardedMPA, ShardedPSO, ShardedGA
    g, s, t = gio.grid("fig7")
    eng = pathfit.Engine(g, device=0)
    comm = Comm(dist, None)
    out = {}
    if what == "maaco":
        sm = ShardedMAACO(comm, lambda: pathfit.MAACO(g, 21, 5, engine=eng, seed=7, **MK), 21, strict=True, chunks=3)
        path, length, turns = sm.solve_path_planning()
        out = dict(path=np.array(path), length=length, turns=turns, tau=sm.local.pheromone_matrix,
                   curve=np.array(sm.local.convergence_curve_data, float))
    elif what == "mpa":
        sm = ShardedMPA(comm, lambda n: pathfit.MPA(g, 30, 9, engine=eng, seed=3, n_local=n), 30)
        res = sm.solve_path_planning()
        out = dict(path=np.array(res[0]), stats=np.array(res[1:], float), curve=np.array(sm.local.convergence_curve_data, float))
    elif what == "pso":
        ps = ShardedPSO(comm, g, engine=eng, seed=6, **PSO_KW)
        # solve() = begin, sweeps, finish -- with what a monitoring caller does in between: fetch_gbest() (a collective) twice in a
        # row and once more inside finish(), with no sweep moving the gbest in between.  Every rank must take the same path through
        # the broadcasts (the owner stays the owner; a second fetch is a no-op everywhere), or the ranks hang or disagree.
        assert ps.begin()
        for _ in range(ps.num_iterations):
            ps.sweep()
        ps.fetch_gbest(); ps.fetch_gbest()
        mid = ps.gbest_particle_data
        assert mid["path"] is not None
        mid_fit, mid_path = mid["fitness"], np.array(mid["path"].tolist() if hasattr(mid["path"], "tolist") else mid["path"])
        res = ps.finish()
        assert res[5] == mid_fit and np.array_equal(np.array(res[0]), mid_path)
        out = dict(path=np.array(res[0]), stats=np.array(res[1:], float), curve=np.array(ps.convergence_curve), pos=ps._pos,
                   pbf=ps._pbest_fit)
    elif what == "ga":
        ga = ShardedGA(comm, g, engine=eng, seed=4, **GA_KW)
        res = ga.solve()
        out = dict(path=np.array(res[0]), stats=np.array(res[1:], float), curve=np.array(ga.convergence_curve),
                   popfit=np.array([p["fitness"] for p in ga.population]))
    np.savez(os.path.join(out_dir, f"{what}{rank}.npz"), **out)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("what", ["maaco", "mpa", "pso", "ga"])
def test_two_ranks_equal_one(tmp_path, what):
    import torch.multiprocessing as mp
    import pathfit
    g, s, t = gio.grid("fig7")
    port = 30100 + os.getpid() % 800 + {"maaco": 0, "mpa": 1, "pso": 2, "ga": 3}[what]
    mp.spawn(_worker, args=(2, port, str(tmp_path), what), nprocs=2, join=True)
    z = [np.load(tmp_path / f"{what}{r}.npz") for r in (0, 1)]
    if what == "maaco":
        m = pathfit.MAACO(g, 21, 5, seed=7, **MK)
        path, length, turns = m.solve_path_planning()
        for zz in z:
            assert np.array_equal(zz["path"], np.array(path)) and float(zz["length"]) == length and float(zz["turns"]) == turns
            assert np.array_equal(zz["tau"], m.pheromone_matrix)               # ordered, pipelined fold == sequential deposits
            assert np.array_equal(zz["curve"], np.array(m.convergence_curve_data, float))
    elif what == "mpa":
        m = pathfit.MPA(g, 30, 9, seed=3)
        res = m.solve_path_planning()
        for zz in z:
            assert np.array_equal(zz["path"], np.array(res[0])) and np.array_equal(zz["stats"], np.array(res[1:], float))
            assert np.array_equal(zz["curve"], np.array(m.convergence_curve_data, float))
    elif what == "pso":
        ps = pathfit.PSOSolver(g, seed=6, **PSO_KW)
        res = ps.solve()
        for zz in z:
            assert np.array_equal(zz["path"], np.array(res[0])) and np.array_equal(zz["stats"], np.array(res[1:], float))
            assert np.array_equal(zz["curve"], np.array(ps.convergence_curve))
        assert np.array_equal(np.concatenate([z[0]["pos"], z[1]["pos"]]), ps._pos)
        assert np.array_equal(np.concatenate([z[0]["pbf"], z[1]["pbf"]]), ps._pbest_fit)
    else:
        ga = pathfit.GASolver(g, seed=4, **GA_KW)
        res = ga.solve()
        for zz in z:
            assert np.array_equal(zz["path"], np.array(res[0])) and np.array_equal(zz["stats"], np.array(res[1:], float))
            assert np.array_equal(zz["curve"], np.array(ga.convergence_curve))
            assert np.array_equal(zz["popfit"], np.array([p["fitness"] for p in ga.population]))


def test_wide_safety_windows_and_dynamic_maps():
    """helper.calculate_path_safety_penalty (helper.py:67-80) accepts any min_safe_distance -- the device
    byte window widens to radius 15 on demand (up to 15.9); beyond that an exact i32 squared-distance transform of the
    needed radius is built on the device (two separable passes) -- and a handle's map can be replaced in place
    (pf_update_grid re-runs the grid preparation on the device)."""
    import pathfit, pf_oracle as po
    from pathfit.engine import score_params, PathfitError
    g, s, t = gio.grid("g256")
    e, o = pathfit.Engine(g), po.Oracle(g)
    rnd = np.random.default_rng(3)
    free = np.flatnonzero(g.reshape(-1) != 1)
    paths = [o.astar(int(a), int(b), None, 0)[0] for a, b in zip(rnd.choice(free, 12), rnd.choice(free, 12))]
    paths = [p for p in paths if len(p) > 1]
    for ms in (1.8, 3.2, 7.0, 7.5, 9.25, 15.9, 16.5, 20.0, 40.0, 20.0, 300.0, 9.25):
        got = e.score_host(paths, score_params(0, True, 0.3, 0.8, ms, 100.0))
        for p, row in zip(paths, got):
            assert np.array_equal(row, o.score(p, 0, 0.3, 0.8, ms, True, 100.0)), ms
    for p, row in zip(paths[:3], e.score_host(paths[:3], score_params(0, True, 0.3, 0.8, 40.0, 100.0))):
        assert np.array_equal(row, o.score(p, 0, 0.3, 0.8, 40.0, True, 100.0, literal_safety=True))      # the O(L * n_obst) scan itself
    with pytest.raises(PathfitError):
        e.score_host(paths, score_params(0, True, 0.3, 0.8, float("inf"), 100.0))
    # a new map in the same handle: searches and scores follow it
    g2 = g.copy(); g2[g2 > 1] = 0
    g2[100:140, 60:200] = 1; g2[0, 0] = 2; g2[-1, -1] = 3
    e.update_grid(g2)
    o2 = po.Oracle(g2)
    free2 = np.flatnonzero(g2.reshape(-1) != 1)
    st_, tg_ = rnd.choice(free2, 16), rnd.choice(free2, 16)
    for variant in (0, 1):
        got, status = e.astar_host(variant, st_, tg_, path_cap=4096)
        for i in range(16):
            assert np.array_equal(got[i], o2.astar(int(st_[i]), int(tg_[i]), None, variant)[0]), (variant, i)
    p2 = [p for p in got if len(p) > 1]
    for ms in (9.25, 25.0):
        sc = e.score_host(p2, score_params(0, True, 0.3, 0.8, ms, 100.0))
        for p, row in zip(p2, sc):
            assert np.array_equal(row, o2.score(p, 0, 0.3, 0.8, ms, True, 100.0))


def test_sharded_solvers_over_rccl_loopback():
    """The rccl transport of pathfit.dist (device pointers, byte counts, stream ordering) on the one GPU of the box: a
    one-rank communicator in loopback mode still issues every all_gather / broadcast / all_reduce through pf_comm_*.  Results
    must equal the plain single-process solvers.  (Multi-rank RCCL itself needs one GPU per rank.)"""
    import pathfit
    from pathfit.dist import Comm, ShardedMAACO, ShardedMPA, ShardedPSO, ShardedGA
    g, s, t = gio.grid("fig7")

    def comm_for(eng):
        return Comm(None, None, engine=eng, transport="rccl", loopback=True)
    # MAACO: gathers, best scan, best-path broadcast, chunked deposit (+ the non-strict all_reduce variant)
    for strict in (True, False):
        e1 = pathfit.Engine(g)
        c1 = comm_for(e1)
        sm = ShardedMAACO(c1, lambda: pathfit.MAACO(g, 21, 5, engine=e1, seed=7, **MK), 21, strict=strict, chunks=3)
        got = sm.solve_path_planning()
        m = pathfit.MAACO(g, 21, 5, seed=7, **MK)
        want = m.solve_path_planning()
        assert got == want and np.array_equal(sm.local.pheromone_matrix, m.pheromone_matrix), strict
        assert c1.calls > 0 and c1.bytes_moved > 0
    e2 = pathfit.Engine(g)
    sm = ShardedMPA(comm_for(e2), lambda n: pathfit.MPA(g, 30, 9, engine=e2, seed=3, n_local=n), 30)
    m = pathfit.MPA(g, 30, 9, seed=3)
    assert sm.solve_path_planning() == m.solve_path_planning()
    assert sm.local.convergence_curve_data == m.convergence_curve_data
    e3 = pathfit.Engine(g)
    ps, ps0 = ShardedPSO(comm_for(e3), g, engine=e3, seed=6, **PSO_KW), pathfit.PSOSolver(g, seed=6, **PSO_KW)
    assert ps.solve() == ps0.solve() and ps.convergence_curve == ps0.convergence_curve
    e4 = pathfit.Engine(g)
    ga, ga0 = ShardedGA(comm_for(e4), g, engine=e4, seed=4, **GA_KW), pathfit.GASolver(g, seed=4, **GA_KW)
    assert ga.solve() == ga0.solve() and ga.convergence_curve == ga0.convergence_curve


def test_head_of_queue_settling_leaves_decodes_unchanged():
    """Default mode (`astar_settle` -1): which decode searches try the parallel settling engine is a POLICY -- the agents at the
    head of the longest-first queue (`astar_settle_top`, per mille of the batch) and / or every search that starts once the
    batch's unfinished agents no longer fill `astar_settle_tail` per mille of the search slots.  Whatever the shares, paths,
    statuses and scores are those of the all-sequential batch."""
    import pathfit
    g = gio.grid("g256")[0]
    e = pathfit.Engine(g)
    n, W, cap = 300, 4, 8 * 256 + 64
    rng = np.random.default_rng(5)
    free = np.flatnonzero(g.reshape(-1) != 1)
    d_wp = e.put(rng.choice(free, (n, W)).astype(np.int32).reshape(-1))
    sp = pathfit.score_params(0, True, 0.3, 0.8, 1.8, 100.0)
    outs = []
    try:
        # (mode, head share, tail share): the tail share is measured against the chip's 2048 search slots, so 300 agents are
        # "the tail" from the start for any share >= 150
        for mode, top, tail in ((0, 0, 0), (-1, 0, 0), (-1, 60, 0), (-1, 500, 0), (-1, 1000, 0), (1, 0, 0), (-1, 0, 20), (-1, 0, 600)):
            e.set_option("astar_settle", mode); e.set_option("astar_settle_top", top); e.set_option("astar_settle_tail", tail)
            dc, dl, ds, dst = e.buf((n, cap), np.int32), e.buf(n, np.int32), e.buf(n, np.int32), e.buf((n, 5), np.float64)
            e.decode_batch(n, W, 0, g.size - 1, cap, dc, dl, ds, d_wp, None, sp, dst)
            c = e.counters()
            outs.append((dc.download(), dl.download(), ds.download(), dst.download(), c["settled_searches"] + c["sequential_searches"]))
    finally:
        e.set_option("astar_settle", -1); e.set_option("astar_settle_top", 0); e.set_option("astar_settle_tail", 400)
    ref = outs[0]
    assert ref[4] == 0 and outs[1][4] == 0                    # nothing tries the engine with a share of 0
    tried = [o[4] for o in outs]
    assert 0 < tried[2] < tried[3] < tried[4] == tried[5]     # 6 % < 50 % < everything == `astar_settle` 1
    assert 0 < tried[6] < tried[7] == tried[5]                # the last 40 agents' remaining links < the whole batch (300 <= 60 % of 2048)
    for o in outs[1:]:
        assert np.array_equal(ref[1], o[1]) and np.array_equal(ref[2], o[2]) and np.array_equal(ref[3], o[3])
        for i in range(n):
            assert np.array_equal(ref[0][i, :ref[1][i]], o[0][i, :o[1][i]]), i
    assert (ref[1] > 0).sum() > n // 2

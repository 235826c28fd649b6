"""CPU oracle == the live, unmodified reference on fresh random cases.
Build container only (needs /root/reference); skipped on the GPU box."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def env():
    import ref_harness as rh
    import pf_oracle as po
    rh.install()
    g = rh.mark_grid(np.array(rh.mods()["env"].grid_fig7_layout_data), (0, 0), (19, 19))
    return rh, po, g


def test_astar_random_triples(env):
    rh, po, g = env
    ra, rd, mpa, orc = rh.RefAStar(g), rh.RefDijkstra(g), rh.make_mpa(g), po.Oracle(g)
    rnd = random.Random(2024)
    free = [tuple(x) for x in np.argwhere(g != 1)]
    for t in range(250):
        s, e = rnd.choice(free), rnd.choice(free)
        if t % 7 == 0: e = s
        if t % 11 == 0: s = (0, 4)
        avoid = set(rnd.sample(free, rnd.randint(0, 40))) if t % 3 else None
        ac = [orc.cell(a) for a in avoid] if avoid is not None else None
        pc, _, cnt = ra.solve(s, e, avoid)
        oc, st = orc.astar(orc.cell(s), orc.cell(e), ac, 0)
        assert np.array_equal(pc, oc) and (len(pc) <= 1 or (cnt["pops"], cnt["pushes"]) == (st[0], st[1]))
        pc, _, cnt = rh.mpa_astar(mpa, s, e, avoid)
        oc, st = orc.astar(orc.cell(s), orc.cell(e), ac, 1)
        assert np.array_equal(pc, oc) and (len(pc) <= 1 or (cnt["pops"], cnt["pushes"]) == (st[0], st[1]))
        pc, _, cnt = rd.solve(s, e, avoid)                      # dijkstra.DijkstraSolver.solve
        oc, st = orc.astar(orc.cell(s), orc.cell(e), ac, 2)
        assert np.array_equal(pc, oc) and (len(pc) <= 1 or (cnt["pops"], cnt["pushes"]) == (st[0], st[1]))


def test_decode_score_random(env):
    rh, po, g = env
    W = dict(turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8, diagonal_obstacle_penalty_value=100.0)
    ga, ps, orc = rh.make_ga(g, **W), rh.make_pso(g, **W), po.Oracle(g)
    rnd = random.Random(99)
    free = [tuple(x) for x in np.argwhere(g != 1)]
    for t in range(120):
        chrom = [rnd.choice(free) for _ in range(5)]
        with rh.quiet():
            p = ga._reconstruct_path_from_chromosome(chrom)
            rs = ga._calculate_stats_for_path(p)
        oc, _ = orc.decode(0, 399, [orc.cell(w) for w in chrom])
        assert np.array_equal(rh.to_cells(p, 20), oc)
        assert list(orc.score(oc, 0, 0.3, 0.8, 1.8, True, 100.0)) == [rs[1], rs[2], rs[3], rs[4], rs[5]]
        pos = [[rnd.uniform(-2, 21), rnd.uniform(-2, 21)] for _ in range(5)]
        with rh.quiet():
            p = ps._reconstruct_path_from_position(pos)
        oc, _ = orc.decode(0, 399, orc.pso_round(pos))
        assert np.array_equal(rh.to_cells(p, 20), oc)


@pytest.mark.parametrize("beta", [7.0, 2.0])
def test_maaco_iterations(env, beta):
    rh, po, g = env
    orc = po.Oracle(g)
    ma = rh.make_maaco(g, num_ants=12, num_iterations=10, alpha=1.0, beta=beta, rho=0.1, Q=2.5, a_turn_coef=1.0,
                       wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5, C0_initial_pheromone=0.1)
    P = po.MaacoParams(alpha=1.0, beta=beta, rho=0.1, Q=2.5, a_turn=1.0, wh_max=0.9, wh_min=0.2, k_h=0.9,
                       q0_initial=0.5, C0=0.1, num_iterations=10)
    tau, dist = orc.maaco_init(0, 399, 0.1)
    assert np.array_equal(tau.reshape(20, 20), ma.pheromone_matrix)
    best = float("inf")
    for it in range(1, 5):
        paths, lens = [], []
        for ant in range(12):
            pc, L, T, draws = rh.maaco_walk(ma, it, 31, ant)
            oc, oL, oT, _ = orc.maaco_walk(0, 399, P, tau, dist, it, 31, ant)
            assert np.array_equal(pc, oc) and L == oL and T == oT
            paths.append(oc); lens.append(oL); best = min(best, oL)
        ma.best_path_length_overall = best
        ma._update_pheromone_trails_maaco([(rh.to_rc(p, 20), l, 0) for p, l in zip(paths, lens)], None)
        orc.maaco_update(tau, 0.1, 2.5, paths, lens, best)
        assert np.array_equal(tau.reshape(20, 20), ma.pheromone_matrix)


@pytest.mark.parametrize("beta", [1.5, 2.0])
def test_mpa_rebuild_random(env, beta):
    rh, po, g = env
    from pathfit import rng as pfrng
    orc, mpa = po.Oracle(g), rh.make_mpa(g, levy_beta=beta)
    base = rh.to_cells(mpa.population[0]["path"], 20)
    a1, _, _ = rh.mpa_astar(mpa, (0, 0), (10, 0))
    a2, _, _ = rh.mpa_astar(mpa, (10, 0), (19, 19), set(rh.to_rc(a1[:-1], 20)))
    alt = np.concatenate([a1, a2[1:]])
    rnd = random.Random(8)
    for t in range(150):
        path_c, el_c = (base, alt) if t % 2 else (alt, base)
        idx = rnd.randint(0, len(path_c) - 1)
        is_levy, scale = t % 3 == 0, rnd.choice([0.5, 0.25, 0.05, 5.0, 40.0])
        pc, res, draws = rh.mpa_rebuild(mpa, rh.to_rc(path_c, 20), rh.to_rc(el_c, 20), idx, is_levy, scale, 17, 2, t)
        gR = orc.rng(17, pfrng.DOM_MPA, 2, t)
        oc, _, _, _ = orc.mpa_rebuild(0, 399, path_c, el_c, idx, is_levy, scale, beta, rh.levy_sigma(beta), gR)
        assert np.array_equal(pc, oc) and gR.ctr == draws
        assert list(orc.score(oc, 1, 0.1, 0.05, 1.5, True, 1000.0)) == [res[1], res[2], res[3], res[4], res[5]]

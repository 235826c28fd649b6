"""CPU oracle (oracle/pf_oracle.c) == golden vectors captured from the
unmodified reference.  Runs anywhere (no GPU, no /root/reference)."""
import math

import numpy as np
import pytest

import golden_io as gio
import pf_oracle as po
from pathfit import rng as pfrng

_orc = {}


def orc(name):
    if name not in _orc:
        g, s, t = gio.grid(name)
        _orc[name] = (po.Oracle(g), s, t)
    return _orc[name]


def test_rng_matches_cpython_derivations():
    z = gio.load("rng")
    L = po.lib()
    for i, key in enumerate(z["keys"]):
        k = [int(v) for v in key]
        g = po.Rng(); L.orc_rng_init(g, *k)
        assert [L.orc_rng_next64(g) for _ in range(8)] == [int(v) for v in z[f"k{i}_next64"]]
        g = po.Rng(); L.orc_rng_init(g, *k)
        assert [L.orc_rng_random(g) for _ in range(8)] == list(z[f"k{i}_random"])
        g = po.Rng(); L.orc_rng_init(g, *k)
        got = [L.orc_rng_randint(g, -5, 17) for _ in range(16)] + [L.orc_rng_randint(g, 0, 0) for _ in range(4)] + \
              [L.orc_rng_randint(g, 0, 2 ** 40) for _ in range(4)]
        assert got == [int(v) for v in z[f"k{i}_randint"]] and g.ctr == int(z[f"k{i}_randint_draws"])
        g = po.Rng(); L.orc_rng_init(g, *k)
        got = [L.orc_rng_normalvariate(g, 0, 1) for _ in range(16)] + [L.orc_rng_normalvariate(g, 0, 0.7) for _ in range(4)]
        assert got == list(z[f"k{i}_normal"]) and g.ctr == int(z[f"k{i}_normal_draws"])
        g = po.Rng(); L.orc_rng_init(g, *k)
        assert [L.orc_rng_uniform(g, 0, 2 * math.pi) for _ in range(8)] == list(z[f"k{i}_uniform"])
        g = po.Rng(); L.orc_rng_init(g, *k)
        got = [L.orc_rng_randbelow(g, n) for n in (1, 2, 3, 5, 8, 100, 1000, 7, 1, 1)]
        assert got == [int(v) for v in z[f"k{i}_choice"]] and g.ctr == int(z[f"k{i}_choice_draws"])
        # python twin agrees with itself across re-keying
        r = pfrng.AgentRandom(*k)
        assert [r.next64() for _ in range(8)] == [int(v) for v in z[f"k{i}_next64"]]


def test_astar_both_variants():
    z = gio.load("astar_cases")
    names = [str(s) for s in z["grid_names"]]
    n = len(z["start"])
    assert n > 400
    for i in range(n):
        o, _, _ = orc(names[int(z["grid_id"][i])])
        avoid = gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None
        path, st = o.astar(int(z["start"][i]), int(z["target"][i]), avoid, int(z["variant"][i]))
        want = gio.csr_get(z["path_off"], z["path"], i)
        assert np.array_equal(path, want), (i, names[int(z["grid_id"][i])])
        if len(want) != 1 and not (len(want) == 0 and z["pops"][i] == 0):
            assert st[0] == z["pops"][i] and st[1] == z["pushes"][i], i


def test_big_cases_at_bench_sizes():
    """The oracle against vectors captured from the UNMODIFIED reference on the bench grids themselves (G512, G1024:
    oracle/capture_golden_big.py): both connectors incl. the corner-to-corner searches the sweeps end on, a path-prefix
    avoid set, heap pop / push counts; GA chained decodes + stats on G512."""
    import pf_oracle as po
    z = gio.load("big_cases")
    names = [str(s) for s in z["grid_names"]]
    orcs = {}
    for i in range(len(z["start"])):
        name = names[int(z["grid_id"][i])]
        if name not in orcs:
            orcs[name] = po.Oracle(gio.upsample(gio.grid("g256")[0], int(name[1:]) // 256))
        avoid = gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None
        path, st = orcs[name].astar(int(z["start"][i]), int(z["target"][i]), avoid, int(z["variant"][i]))
        want = gio.csr_get(z["path_off"], z["path"], i)
        assert np.array_equal(path, want), (i, name)
        if len(want) > 1:
            assert st[0] == z["pops"][i] and st[1] == z["pushes"][i], (i, name, st[:2], z["pops"][i], z["pushes"][i])
    assert max(int(v) for v in z["pops"]) > 90000                       # the 512^2 corner-to-corner search is in there
    o = orcs["G512"]
    for j in range(len(z["dec_wp"])):
        wp = z["dec_wp"][j]; wp = wp[wp >= 0]                                # (3 or 5 waypoints)
        p, _ = o.decode(0, 512 * 512 - 1, wp)
        assert np.array_equal(p, gio.csr_get(z["dec_path_off"], z["dec_path"], j))
        assert np.array_equal(o.score(p, 0, 0.3, 0.8, 1.8, True, 100.0), z["dec_stats"][j])
    # MPA._reconstruct_path_segment of the reference's own initial path at 512^2 (main.py:44-52 parameters)
    base = z["reb_base"]
    seed, it = (int(v) for v in z["reb_seed_it"])
    assert np.array_equal(o.score(base, 1, 0.1, 0.8, 1.8, True, 100.0), z["reb_base_stats"])
    for i in range(len(z["reb_idx"])):
        g = o.rng(seed, pfrng.DOM_MPA, it, int(z["reb_agent"][i]))
        out, _, _, _ = o.mpa_rebuild(0, 512 * 512 - 1, base, base, int(z["reb_idx"][i]), int(z["reb_is_levy"][i]), float(z["reb_scale"][i]),
                                     2.0, float(z["reb_sigma"][0]), g)
        assert np.array_equal(out, gio.csr_get(z["reb_out_off"], z["reb_out"], i)) and g.ctr == z["reb_draws"][i], i
        assert np.array_equal(o.score(out, 1, 0.1, 0.8, 1.8, True, 100.0), z["reb_stats"][i]), i


def test_big_maaco_walks_and_pheromone_at_bench_sizes():
    """MAACO.py:278-332 of the unmodified reference on G512 (8 ants x 2 iterations, beta 7 and beta 2) and G1024 (4 ants):
    every ant's walk, length and turn count, and the whole pheromone matrix after every update."""
    z = gio.load("big_cases")
    bp = z["maaco_base_params"]
    branches = np.zeros(3, np.int64)
    for ri in range(int(z["maaco_runs"])):
        R, beta, n_ants, n_it, K, seed = z[f"maaco{ri}_cfg"]
        R, n_ants, n_it, K, seed = int(R), int(n_ants), int(n_it), int(K), int(seed)
        o = po.Oracle(gio.upsample(gio.grid("g256")[0], R // 256))
        s, t = 0, R * R - 1
        P = po.MaacoParams(alpha=bp[0], beta=beta, rho=bp[1], Q=bp[2], a_turn=bp[3], wh_max=bp[4], wh_min=bp[5],
                           k_h=bp[6], q0_initial=bp[7], C0=bp[8], num_iterations=K)
        tau, dist = o.maaco_init(s, t, bp[8])
        best = float("inf"); k = 0
        for it in range(1, n_it + 1):
            paths, lens = [], []
            for ant in range(n_ants):
                p, L, T, cnt = o.maaco_walk(s, t, P, tau, dist, it, seed, ant)
                assert np.array_equal(p, gio.csr_get(z[f"maaco{ri}_path_off"], z[f"maaco{ri}_path"], k)), (ri, it, ant)
                assert L == z[f"maaco{ri}_len"][k] and (T if T != float("inf") else -1) == z[f"maaco{ri}_turns"][k], (ri, it, ant)
                branches += cnt[2:5]
                paths.append(p); lens.append(L); best = min(best, L); k += 1
            o.maaco_update(tau, bp[1], bp[2], paths, lens, best)
            assert np.array_equal(tau.reshape(R, R), z[f"maaco{ri}_tau"][it - 1]), (ri, it)
    assert branches[0] > 1000 and branches[1] + branches[2] > 1000      # greedy and non-greedy steps both occur (MAACO.py:241 / :251)


def test_dijkstra_solver():
    """DijkstraSolver.solve (dijkstra.py:32-97) = variant 2 of the restated connector."""
    z = gio.load("dijkstra_cases")
    names = [str(s) for s in z["grid_names"]]
    n = len(z["start"])
    assert n > 100
    for i in range(n):
        o, _, _ = orc(names[int(z["grid_id"][i])])
        avoid = gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None
        path, st = o.astar(int(z["start"][i]), int(z["target"][i]), avoid, 2)
        want = gio.csr_get(z["path_off"], z["path"], i)
        assert np.array_equal(path, want), (i, names[int(z["grid_id"][i])])
        if len(want) != 1 and not (len(want) == 0 and z["pops"][i] == 0):
            assert st[0] == z["pops"][i] and st[1] == z["pushes"][i], i


def test_decode_and_score():
    z = gio.load("decode_cases")
    names = [str(s) for s in z["grid_names"]]
    for i in range(len(z["kind"])):
        o, s, t = orc(names[int(z["grid_id"][i])])
        w = z["main_w"] if str(z["weights"][i]) == "main" else z["def_w"]
        wp = gio.csr_get(z["wp_off"], z["wp"], i)
        want = gio.csr_get(z["path_off"], z["path"], i)
        kind = int(z["kind"][i])
        if kind == 0:
            path, _ = o.decode(s, t, wp.astype(np.int32))
        elif kind == 1:
            path, _ = o.decode(s, t, o.pso_round(wp))
        else:
            path = want
        assert np.array_equal(path, want), i
        sc = o.score(path, 0, w[0], w[1], w[2], True, w[3])
        sc_lit = o.score(path, 0, w[0], w[1], w[2], True, w[3], literal_safety=True)
        assert np.array_equal(sc, z["stats"][i]) and np.array_equal(sc_lit, sc), (i, sc, z["stats"][i])
    assert (z["stats"][:, 3] > 0).any()      # the corner-cutting path exercises diag


def test_maaco_walks_and_pheromone():
    z = gio.load("maaco_cases")
    bp = z["base_params"]
    for ri, gname in enumerate(z["runs_grid"]):
        beta, n_ants, n_it, K, seed = z["runs_num"][ri]
        n_ants, n_it, K, seed = int(n_ants), int(n_it), int(K), int(seed)
        o, s, t = orc(str(gname))
        P = po.MaacoParams(alpha=bp[0], beta=beta, rho=bp[1], Q=bp[2], a_turn=bp[3], wh_max=bp[4], wh_min=bp[5],
                           k_h=bp[6], q0_initial=bp[7], C0=bp[8], num_iterations=K)
        tau, dist = o.maaco_init(s, t, bp[8])
        assert np.array_equal(tau.reshape(o.R, o.C), z[f"r{ri}_tau0"])
        best = float("inf"); k = 0
        for it in range(1, n_it + 1):
            paths, lens = [], []
            for ant in range(n_ants):
                p, L, T, _ = o.maaco_walk(s, t, P, tau, dist, it, seed, ant)
                assert np.array_equal(p, gio.csr_get(z[f"r{ri}_path_off"], z[f"r{ri}_path"], k)), (ri, it, ant)
                assert L == z[f"r{ri}_len"][k] and (T if T != float("inf") else -1) == z[f"r{ri}_turns"][k]
                paths.append(p); lens.append(L); best = min(best, L); k += 1
            o.maaco_update(tau, bp[1], bp[2], paths, lens, best)
            if f"r{ri}_tau" in z:
                assert np.array_equal(tau.reshape(o.R, o.C), z[f"r{ri}_tau"][it - 1]), (ri, it)
        if f"r{ri}_tau_sum" in z:
            m = tau.reshape(o.R, o.C)
            assert np.array_equal(m[:4], z[f"r{ri}_tau_last_rows"])
            assert np.array_equal(np.array([m.sum(), m.max(), m.min()]), z[f"r{ri}_tau_sum"])
    for K in (3, 50, 100):
        got = np.array([po.lib().orc_maaco_q0(i, K, 0.5) for i in range(1, K + 1)])
        assert np.array_equal(got, z[f"q0_K{K}"])


def test_mpa_rebuild():
    z = gio.load("mpa_cases")
    names = [str(s) for s in z["grid_names"]]
    seed, it = (int(v) for v in z["seed_it"])
    changed = 0
    for i in range(len(z["idx"])):
        o, s, t = orc(names[int(z["grid_id"][i])])
        beta = float(z["beta"][i]); sigma = float(z["sigma"][0 if beta == 1.5 else 1])
        g = o.rng(seed, pfrng.DOM_MPA, it, int(z["agent"][i]))
        inp = gio.csr_get(z["in_off"], z["in_path"], i); el = gio.csr_get(z["el_off"], z["el_path"], i)
        out, isnew, tcell, _ = o.mpa_rebuild(s, t, inp, el, int(z["idx"][i]), int(z["is_levy"][i]),
                                             float(z["scale"][i]), beta, sigma, g)
        want = gio.csr_get(z["out_off"], z["out_path"], i)
        assert np.array_equal(out, want) and g.ctr == z["draws"][i], i
        sc = o.score(out, 1, 0.1, 0.05, 1.5, True, 1000.0)
        assert np.array_equal(sc, z["stats"][i]), i
        changed += not np.array_equal(out, inp)
    assert changed > 50


def test_pso_update():
    z = gio.load("pso_update")
    g, s, t = gio.grid("fig7")
    o = po.Oracle(g)
    for i in range(len(z["seed"])):
        w, c1, c2, mv = z["params"][i]
        pos, vel = o.pso_update(z["pos0"][i][None], z["vel0"][i][None], z["pbest"][i][None], z["gbest"][i],
                                w, c1, c2, mv, int(z["seed"][i]), int(z["it"][i]), 0)
        assert np.array_equal(pos[0], z["pos1"][i]) and np.array_equal(vel[0], z["vel1"][i]), i

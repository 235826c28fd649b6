"""Full solver loops == the UNMODIFIED reference's own solve loops run end to end under the per-agent stream
contract (tests/golden/e2e.npz, captured by oracle/ref_e2e.py).  CPU part: the oracle-driven loops and the
host logic of the GA facade; GPU part (marked): the shipped facades."""
import numpy as np
import pytest

import golden_io as gio

MK = dict(alpha=1.0, rho=0.1, Q=2.5, a_turn_coef=1.0, wh_max=0.9, wh_min=0.2, k_h_adaptive=0.9, q0_initial=0.5)
MPA_MAIN = dict(FADs_rate=0.2, P_const=0.5, levy_beta=2.0, turn_penalty_factor=0.1, safety_penalty_factor=0.8,
                min_safe_distance=1.8, diagonal_obstacle_penalty=100.0)
GA_KW = dict(num_generations=6, population_size=24, num_waypoints_per_chromosome=5, mutation_rate=0.1, crossover_rate=0.8,
             tournament_size=3, turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
             diagonal_obstacle_penalty_value=100.0)


def curve_eq(a, b):
    a = np.array([np.nan if v is None else v for v in a], float)
    return np.array_equal(a, np.asarray(b, float), equal_nan=True)


def test_oracle_loops_match_reference_solve_loops():
    import pf_loops, pf_oracle as po
    z = gio.load("e2e")
    for i in range(3):
        g, s, t = gio.grid(str(z[f"maaco{i}_grid"]))
        beta, ants, iters, seed = z[f"maaco{i}_cfg"]
        r = pf_loops.maaco_solve(po.Oracle(g), s, t, int(ants), int(iters), beta=float(beta), C0=0.1, seed=int(seed), **MK)
        assert np.array_equal(r["path"], z[f"maaco{i}_path"]) and r["length"] == z[f"maaco{i}_length"]
        assert r["turns"] == z[f"maaco{i}_turns"] and np.array_equal(r["tau"], z[f"maaco{i}_tau"])
        assert curve_eq(r["curve"], z[f"maaco{i}_curve"])
    for i in range(4):
        g, s, t = gio.grid(str(z[f"mpa{i}_grid"]))
        seed, n, it, main = (int(v) for v in z[f"mpa{i}_cfg"])
        kw = dict(FADs_rate=0.2, P_const=0.5, levy_beta=2.0, w_turn=0.1, w_safe=0.8, min_safe=1.8, diag_pen=100.0) if main else {}
        ref = pf_loops.MpaOracle(po.Oracle(g), s, t, n, it, seed=seed, **kw)
        best = ref.solve()
        assert np.array_equal(best[0], z[f"mpa{i}_path"]) and np.array_equal(best[1], z[f"mpa{i}_stats"]), i
        assert curve_eq(ref.curve, z[f"mpa{i}_curve"])
        assert np.array_equal([p[1][4] for p in ref.pop], z[f"mpa{i}_pop_fitness"])


class _NoEngine:
    """The GA host logic needs no device when decode+score come from the oracle (checker only)."""


def _oracle_backed_ga(orc):
    import pathfit
    from pathfit.paths import CellPath

    class OB(pathfit.GASolver):
        def _evaluate(self, wp_cells=None, wp_pos=None):
            n = len(wp_cells)
            cps, stats, feas = [], np.zeros((n, 5)), np.zeros(n, bool)
            for i in range(n):
                p, _ = orc.decode(self._cell(self.start_node), self._cell(self.target_node), wp_cells[i])
                sp = self._sp
                stats[i] = orc.score(p, 0, sp.w_turn, sp.w_safe, sp.min_safe, bool(sp.restrict_policy), sp.diag_pen)
                cps.append(CellPath(p, self.cols)); feas[i] = len(p) > 0
            return cps, stats, feas
    return OB


def test_ga_host_logic_matches_reference_solve():
    import pf_oracle as po
    z = gio.load("e2e")
    g, s, t = gio.grid("fig7")
    ga = _oracle_backed_ga(po.Oracle(g))(g, engine=_NoEngine(), seed=4, **GA_KW)
    res = ga.solve()
    assert [r * 20 + c for r, c in res[0]] == list(z["ga0_path"])
    assert np.array_equal(np.array(res[1:], float), z["ga0_stats"])
    assert np.array_equal(np.array(ga.convergence_curve), z["ga0_curve"])
    assert np.array_equal([p["fitness"] for p in ga.population], z["ga0_pop_fitness"])


@pytest.mark.gpu
def test_gpu_facades_match_reference_solve_loops():
    import pathfit
    z = gio.load("e2e")
    for i in range(3):
        g, s, t = gio.grid(str(z[f"maaco{i}_grid"]))
        beta, ants, iters, seed = z[f"maaco{i}_cfg"]
        m = pathfit.MAACO(g, int(ants), int(iters), beta=float(beta), C0_initial_pheromone=0.1, seed=int(seed), **MK)
        path, length, turns = m.solve_path_planning()
        assert [r * m.cols + c for r, c in path] == list(z[f"maaco{i}_path"]) and length == z[f"maaco{i}_length"]
        assert turns == z[f"maaco{i}_turns"] and np.array_equal(m.pheromone_matrix, z[f"maaco{i}_tau"])
        assert curve_eq(m.convergence_curve_data, z[f"maaco{i}_curve"])
    for i in range(4):
        g, s, t = gio.grid(str(z[f"mpa{i}_grid"]))
        seed, n, it, main = (int(v) for v in z[f"mpa{i}_cfg"])
        m = pathfit.MPA(g, n, it, seed=seed, **(MPA_MAIN if main else {}))
        res = m.solve_path_planning()
        assert [r * m.cols + c for r, c in res[0]] == list(z[f"mpa{i}_path"]), i
        assert np.array_equal(np.array(res[1:], float), z[f"mpa{i}_stats"]) and curve_eq(m.convergence_curve_data, z[f"mpa{i}_curve"])
        assert np.array_equal([p["fitness"] for p in m.population], z[f"mpa{i}_pop_fitness"])
    g, s, t = gio.grid("fig7")
    ga = pathfit.GASolver(g, seed=4, **GA_KW)
    res = ga.solve()
    assert [r * 20 + c for r, c in res[0]] == list(z["ga0_path"]) and np.array_equal(np.array(res[1:], float), z["ga0_stats"])
    assert np.array_equal(np.array(ga.convergence_curve), z["ga0_curve"])
    # PSO with the reference's asynchronous gbest (pso.py:222-229) reproduced by speculate-and-repair
    ps = pathfit.PSOSolver(g, num_iterations=8, num_particles=24, num_waypoints_per_particle=5, w=0.7, c1=1.5, c2=1.5,
                           turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
                           diagonal_obstacle_penalty_value=100.0, seed=6)
    res = ps.solve()
    assert [r * 20 + c for r, c in res[0]] == list(z["pso0_path"]) and np.array_equal(np.array(res[1:], float), z["pso0_stats"])
    assert np.array_equal(np.array(ps.convergence_curve), z["pso0_curve"])
    assert np.array_equal(ps._pos, z["pso0_pos"]) and np.array_equal(ps._pbest_fit, z["pso0_pbest_fit"])


@pytest.mark.gpu
def test_gpu_pso_init_fallback_matches_reference():
    """pso.py:126-143: when none of the 20 N random particles decodes (a serpentine corridor map), the reference takes the
    direct A* path as its one particle, clones it and iterates; the facade must do the same."""
    import pathfit
    z = gio.load("e2e_pso_fallback")
    g = z["grid"].astype(np.int64)
    ps = pathfit.PSOSolver(g, num_iterations=4, num_particles=4, num_waypoints_per_particle=5, w=0.7, c1=1.5, c2=1.5,
                           turn_penalty_factor=0.3, safety_penalty_factor=0.8, min_safe_distance=1.8,
                           diagonal_obstacle_penalty_value=100.0, seed=21)
    res = ps.solve()
    C = g.shape[1]
    assert [r * C + c for r, c in res[0]] == list(z["path"]) and np.array_equal(np.array(res[1:], float), z["stats"])
    assert np.array_equal(np.array(ps.convergence_curve), z["curve"])
    assert np.array_equal(ps._pos, z["pos"]) and np.array_equal(ps._pbest_fit, z["pbest_fit"])

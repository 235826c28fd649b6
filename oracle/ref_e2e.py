"""End-to-end runs of the UNMODIFIED reference solver loops under the per-agent stream contract.

TEST INFRASTRUCTURE (build container only).  No reference file is edited: instance methods are wrapped and
`MPA` is subclassed at run time so that the harness can re-key the shared AgentRandom at every agent
boundary of the reference's own loops:
  MAACO  _construct_ant_solution_maaco(ant, iter)            -> (seed, DOM_MAACO, iter, ant)
  MPA    self.population[i] at the top of each predator body  -> (seed, DOM_MPA, iter, i)
         self.population = [] / .append in the FADs loop      -> (seed, DOM_MPA_FADS, iter, i)
  GA     _create_chromosome -> (seed, DOM_INIT, 0, attempt); _selection -> (seed, DOM_GA_SELECT, gen, 0);
         _crossover (+ the two _mutate calls that follow)     -> (seed, DOM_GA, gen, pair)
"""
import numpy as np

import ref_harness as rh
from pathfit import rng as pfrng


def maaco_solve(grid, seed, **params):
    ma = rh.make_maaco(grid, **params)
    orig = ma._construct_ant_solution_maaco

    def hooked(ant_id, it):
        rh.RNG.rekey(seed, pfrng.DOM_MAACO, it, ant_id)
        return orig(ant_id, it)
    ma._construct_ant_solution_maaco = hooked
    with rh.quiet():
        path, length, turns = ma.solve_path_planning()
    return dict(path=rh.to_cells(path, ma.cols), length=length, turns=turns,
                curve=np.array([np.nan if v is None else v for v in ma.convergence_curve_data]),
                tau=ma.pheromone_matrix.copy())


def mpa_solve(grid, seed, num_predators, num_iterations, **kw):
    m = rh.install()
    MPAref = m["MPA"].MPA
    st = dict(sorts=0, it=0, fads=False)

    class HookList(list):
        def __getitem__(self, i):
            if isinstance(i, int) and not st["fads"] and st["it"] > 0:
                rh.RNG.rekey(seed, pfrng.DOM_MPA, st["it"], i)
            return list.__getitem__(self, i)

        def sort(self, *a, **k):
            st["sorts"] += 1
            if st["sorts"] >= 2 and st["sorts"] % 2 == 0:      # :333 -> a new iteration starts
                st["it"] = st["sorts"] // 2
            st["fads"] = False
            return list.sort(self, *a, **k)

        def append(self, x):
            list.append(self, x)
            if st["fads"]:
                rh.RNG.rekey(seed, pfrng.DOM_MPA_FADS, st["it"], len(self))

    class Hooked(MPAref):
        @property
        def population(self):
            return self.__dict__["_pop"]

        @population.setter
        def population(self, v):
            hl = HookList(v)
            self.__dict__["_pop"] = hl
            if st["it"] > 0 and len(hl) == 0:                    # :386 `self.population = []` opens the FADs loop
                st["fads"] = True
                rh.RNG.rekey(seed, pfrng.DOM_MPA_FADS, st["it"], 0)

    rh.RNG.rekey(0, 0, 0, 0)
    with rh.quiet():
        mp = Hooked(np.array(grid), num_predators, num_iterations, **kw)
        res = mp.solve_path_planning()
    C = mp.cols
    return dict(path=rh.to_cells(res[0], C), stats=np.array([res[1], res[2], res[3], res[4], res[5]], float),
                curve=np.array([np.nan if v is None else v for v in mp.convergence_curve_data]),
                pop_fitness=np.array([p["fitness"] for p in mp.population]),
                pop_len=np.array([len(p["path"]) for p in mp.population]))


def ga_solve(grid, seed, **kw):
    m = rh.install()
    with rh.quiet():
        ga = m["ga_solver"].GASolver(np.array(grid), **kw)
    st = dict(attempt=0, gen=-1, pair=0)
    o_create, o_sel, o_cross = ga._create_chromosome, ga._selection, ga._crossover

    def create():
        rh.RNG.rekey(seed, pfrng.DOM_INIT, 0, st["attempt"]); st["attempt"] += 1
        return o_create()

    def sel():
        st["gen"] += 1; st["pair"] = 0
        rh.RNG.rekey(seed, pfrng.DOM_GA_SELECT, st["gen"], 0)
        return o_sel()

    def cross(a, b):
        rh.RNG.rekey(seed, pfrng.DOM_GA, st["gen"], st["pair"]); st["pair"] += 1
        return o_cross(a, b)
    ga._create_chromosome, ga._selection, ga._crossover = create, sel, cross
    with rh.quiet():
        res = ga.solve()
    return dict(path=rh.to_cells(res[0], ga.cols), stats=np.array([res[1], res[2], res[3], res[4], res[5]], float),
                curve=np.array(ga.convergence_curve, float), attempts=st["attempt"],
                pop_fitness=np.array([p["fitness"] for p in ga.population]))


def pso_solve(grid, seed, **kw):
    """PSOSolver.solve() of the unmodified reference (asynchronous gbest, pso.py:178-229).  Streams: init attempt k
    -> (seed, DOM_INIT, 0, k) (re-keyed at the first _generate_random_waypoint call of each attempt); particle p of
    iteration it -> (seed, DOM_PSO, it, p) (re-keyed right after the previous particle's _reconstruct call)."""
    m = rh.install()
    with rh.quiet():
        ps = m["pso"].PSOSolver(np.array(grid), **kw)
    W, N = ps.num_waypoints, ps.num_particles
    st = dict(wp=0, rec=0, run=False)
    o_wp, o_rec, o_init = ps._generate_random_waypoint, ps._reconstruct_path_from_position, ps._initialize_particles

    def wp():
        if st["wp"] % W == 0:
            rh.RNG.rekey(seed, pfrng.DOM_INIT, 0, st["wp"] // W)
        st["wp"] += 1
        return o_wp()

    def rec(pos):
        r = o_rec(pos)
        if st["run"]:
            st["rec"] += 1
            it, p = divmod(st["rec"], N)
            rh.RNG.rekey(seed, pfrng.DOM_PSO, it, p)
        return r

    def init():
        ok = o_init()
        st["run"] = True
        rh.RNG.rekey(seed, pfrng.DOM_PSO, 0, 0)
        return ok
    ps._generate_random_waypoint, ps._reconstruct_path_from_position, ps._initialize_particles = wp, rec, init
    with rh.quiet():
        res = ps.solve()
    return dict(path=rh.to_cells(res[0], ps.cols), stats=np.array([res[1], res[2], res[3], res[4], res[5]], float),
                curve=np.array(ps.convergence_curve, float), attempts=st["wp"] // W,
                pos=np.array([p["position"] for p in ps.particles], float),
                pbest_fit=np.array([p["pbest_fitness"] for p in ps.particles], float))

"""Oracle-backed stand-ins for pathfit.MAACO / Engine so that the sharding logic of
pathfit/dist.py (exchange steps, global ordering, ordered pheromone fold) can run on CPU ranks."""
import numpy as np

import pf_oracle as po
from pathfit.paths import CellPath

INF = float("inf")


class FakeMaacoEngine:
    def __init__(self, orc, rho, Q):
        self.o, self.rho, self.Q = orc, rho, Q
        self.tau = None
        self.pending = None

    def maaco_evaporate(self):
        self.tau = self.tau * (1.0 - self.rho)

    def maaco_deposit(self, n, cap, dc, dl, dp):
        paths, lens = self.pending
        for p, L in zip(paths, lens):                       # MAACO.py:306-311 in ant order
            if L != INF and len(p) and L > 1e-6:
                self.tau[p] += self.Q / L

    def maaco_clip(self, best_len):
        R, C = self.o.R, self.o.C
        bl = float(R + C) if best_len == INF else best_len
        bl = max(bl, 1e-6)
        tmax = (1.0 / (1.0 - self.rho)) * (1.0 / bl)
        tmin = tmax / (2.0 * max(C, R, 1))
        occ = self.o.occ.reshape(-1) == 1
        self.tau = np.where(occ, 1e-9, np.clip(self.tau, tmin, tmax))

    def maaco_get_pheromone(self):
        return self.tau.reshape(self.o.R, self.o.C).copy()

    def maaco_set_pheromone(self, t):
        self.tau = np.asarray(t, np.float64).reshape(-1).copy()


class FakeMAACO:
    """Same attributes/methods ShardedMAACO uses from pathfit.MAACO."""

    def __init__(self, grid, start, target, num_ants, num_iterations, seed, **kw):
        self.o = po.Oracle(grid)
        self.rows, self.cols = self.o.R, self.o.C
        self.s, self.t, self.seed = start, target, seed
        self.num_ants, self.num_iterations = num_ants, num_iterations
        self.P = po.MaacoParams(alpha=kw["alpha"], beta=kw["beta"], rho=kw["rho"], Q=kw["Q"], a_turn=kw["a_turn_coef"],
                                wh_max=kw["wh_max"], wh_min=kw["wh_min"], k_h=kw["k_h_adaptive"], q0_initial=kw["q0_initial"],
                                C0=0.1, num_iterations=num_iterations)
        self.engine = FakeMaacoEngine(self.o, kw["rho"], kw["Q"])
        tau, self.dist = self.o.maaco_init(start, target, 0.1)
        self.engine.tau = tau
        self.best_path_overall, self.best_path_length_overall, self.best_path_turns_overall = [], INF, INF
        self.convergence_curve_data = []
        self.path_cap = 0
        self._bufs = (None, None, None, None)
        self._paths = []

    def walk_iteration(self, it, ant0, n):
        paths, lens, turns = [], [], []
        for a in range(ant0, ant0 + n):
            p, L, T, _ = self.o.maaco_walk(self.s, self.t, self.P, self.engine.tau, self.dist, it, self.seed, a)
            paths.append(p); lens.append(L); turns.append(-1 if T == INF else int(T))
        self._paths = paths
        self.engine.pending = (paths, lens)
        return np.array(lens), np.array(turns, np.int32)

    def ant_path(self, li):
        return CellPath(self._paths[li], self.cols)

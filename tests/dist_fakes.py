"""Oracle-backed stand-ins for pathfit.MAACO / Engine so that the sharding logic of
pathfit/dist.py (exchange steps, global ordering, ordered pheromone fold) can run on CPU ranks."""
import numpy as np

import pf_oracle as po
from pathfit.paths import CellPath

INF = float("inf")


class FakeMaacoEngine:
    """The slice of pathfit.Engine that ShardedMAACO drives, over numpy (buffers are dist.HostBuf objects)."""

    def __init__(self, orc, rho, Q):
        from pathfit.dist import HostBuf
        self.o, self.rho, self.Q = orc, rho, Q
        self._tau = HostBuf(orc.R * orc.C, np.float64)
        self.pending = None

    @property
    def tau(self):
        return self._tau.a

    @tau.setter
    def tau(self, v):
        self._tau.a[:] = np.asarray(v, np.float64).reshape(-1)

    @property
    def tau_buf(self):
        return self._tau

    def buf(self, shape, dtype):
        from pathfit.dist import HostBuf
        return HostBuf(shape, dtype)

    def maaco_evaporate(self):
        self.tau = self.tau * (1.0 - self.rho)

    def maaco_deposit_begin(self, n, cap, dc, dl, dp):
        pass

    def maaco_deposit_cells(self, c0, c1):
        paths, lens = self.pending
        t = self.tau
        for p, L in zip(paths, lens):                       # MAACO.py:306-311 in ant order, cells of this chunk only
            if L != INF and len(p) and L > 1e-6:
                p = np.asarray(p)
                q = p[(p >= c0) & (p < c1)]
                t[q] += self.Q / L

    def maaco_deposit(self, n, cap, dc, dl, dp):
        self.maaco_deposit_cells(0, self.o.R * self.o.C)

    def maaco_best_dev(self, n, bp, bt):
        from pathfit.dist import maaco_best_scan_host
        return maaco_best_scan_host(bp.a[:n], bt.a[:n])

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
        self.path_cap = 2 * self.rows * self.cols
        self._paths = []
        self._wb = None

    def walk_iteration_dev(self, it, ant0, n):
        from pathfit.dist import HostBuf
        paths, lens, turns = [], [], []
        for a in range(ant0, ant0 + n):
            p, L, T, _ = self.o.maaco_walk(self.s, self.t, self.P, self.engine.tau, self.dist, it, self.seed, a)
            paths.append(p); lens.append(L); turns.append(-1 if T == INF else int(T))
        self._paths = paths
        self.engine.pending = (paths, lens)
        cap = self.path_cap
        dc, dl, dp, dt, ds = HostBuf((max(n, 1), cap), np.int32), HostBuf(max(n, 1), np.int32), HostBuf(max(n, 1), np.float64), \
            HostBuf(max(n, 1), np.int32), HostBuf(max(n, 1), np.int32)
        for i, p in enumerate(paths):
            dc.write(i * cap, p); dl.write(i, [len(p)])
        dp.write(0, lens); dt.write(0, turns)
        self._wb = (dc, dl, dp, dt, ds)
        return n

    def walk_bufs(self):
        return self._wb

    def walk_iteration(self, it, ant0, n):
        self.walk_iteration_dev(it, ant0, n)
        return self._wb[2].a[:n].copy(), self._wb[3].a[:n].copy()

    def ant_path(self, li):
        return CellPath(self._paths[li], self.cols)

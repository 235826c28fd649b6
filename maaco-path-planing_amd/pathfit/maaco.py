"""MAACO with the reference's constructor / solve_path_planning() surface
(MAACO.py:10-14, :334-371).  Ant walks (:278-302) and the pheromone update
(:304-332) run on the GPU; the best-of-iteration bookkeeping (:343-358) stays
on the host, fed by the per-ant length/turn arrays.

Sharding: ants [ant0, ant0+n_local) of every iteration are walked on this
rank's GPU; the per-ant streams are keyed by the GLOBAL ant index, so results
do not depend on the partition.  `pathfit.dist.ShardedMAACO` adds the
exchange step.
"""
import math

import numpy as np

from ._lib import MaacoParams
from .engine import Engine
from .env import START_NODE_VAL, TARGET_NODE_VAL, find_marker
from .paths import CellPath

INF = float("inf")


class MAACO:
    def __init__(self, grid, num_ants, num_iterations, alpha, beta, rho, Q, a_turn_coef, wh_max, wh_min,
                 k_h_adaptive, q0_initial, C0_initial_pheromone=0.1, engine=None, device=0, seed=0, verbose=False):
        self.grid = np.array(grid, dtype=int)
        self.rows, self.cols = self.grid.shape
        self.num_ants, self.num_iterations = num_ants, num_iterations
        self.alpha, self.beta, self.rho, self.Q = alpha, beta, rho, Q
        self.a_turn_coef, self.wh_max, self.wh_min = a_turn_coef, wh_max, wh_min
        self.k_h_adaptive, self.q0_initial = k_h_adaptive, q0_initial
        self.k0_iter_threshold_factor = 0.7
        self.C0_base = C0_initial_pheromone
        self.start_node = find_marker(self.grid, START_NODE_VAL, "MAACO")
        self.target_node = find_marker(self.grid, TARGET_NODE_VAL, "MAACO")
        self.dist_S_to_T_overall = max(math.sqrt((self.start_node[0] - self.target_node[0]) ** 2 +
                                                 (self.start_node[1] - self.target_node[1]) ** 2), 1e-9)
        self.seed, self.verbose = int(seed), verbose
        self.engine = engine if engine is not None else Engine(self.grid, device)
        s = self.start_node[0] * self.cols + self.start_node[1]
        t = self.target_node[0] * self.cols + self.target_node[1]
        self.engine.maaco_setup(MaacoParams(float(alpha), float(beta), float(rho), float(Q), float(a_turn_coef),
                                            float(wh_max), float(wh_min), float(k_h_adaptive), float(q0_initial),
                                            float(C0_initial_pheromone), int(num_iterations), s, t))
        self._best_path, self._best_on_device = [], False   # best_path_overall: host list, or still in HBM (materialised on access)
        self.best_path_length_overall = INF
        self.best_path_turns_overall = INF
        self.convergence_curve_data = []
        self.path_cap = min(self.rows * self.cols, 6 * (self.rows + self.cols) + 64)
        self._bufs = None

    # public state the reference exposes (MAACO.py:47-48, :373)
    @property
    def best_path_overall(self):
        """MAACO.py:351-358: a list of (r, c).  The one-enqueue iteration keeps the row in HBM; it crosses PCIe here, once."""
        if self._best_on_device:
            self._best_path = CellPath(self.engine.maaco_best_path(self.path_cap), self.cols).tolist()
            self._best_on_device = False
        return self._best_path

    @best_path_overall.setter
    def best_path_overall(self, v):
        self._best_path, self._best_on_device = v, False

    @property
    def pheromone_matrix(self):
        return self.engine.maaco_get_pheromone()

    @pheromone_matrix.setter
    def pheromone_matrix(self, tau):
        self.engine.maaco_set_pheromone(tau)

    @property
    def dist_to_target_matrix(self):
        r, c = np.mgrid[0:self.rows, 0:self.cols]
        return np.sqrt(((r - self.target_node[0]) ** 2 + (c - self.target_node[1]) ** 2).astype(np.float64))

    def _alloc(self, n):
        e = self.engine
        if self._bufs is None or self._bufs[0] != (n, self.path_cap):
            self._bufs = ((n, self.path_cap), e.buf((n, self.path_cap), np.int32), e.buf(n, np.int32),
                          e.buf(n, np.float64), e.buf(n, np.int32), e.buf(n, np.int32))
        return self._bufs[1:]

    def walk_iteration_dev(self, iter_num, ant0=0, n=None):
        """Walk ants [ant0, ant0+n) of iteration iter_num; everything stays in HBM (paths, lengths, turns, status:
        self.walk_bufs()).  The only device-to-host traffic is the overflow counter the walk batch already returns."""
        n = self.num_ants if n is None else n
        while True:
            dc, dl, dp, dt, ds = self._alloc(n)
            self.engine.maaco_walk(iter_num, self.seed, ant0, n, self.path_cap, dc, dl, dp, dt, ds)
            if self.engine.counters()["overflow_agents"] and self.path_cap < self.rows * self.cols:
                self.path_cap = min(self.rows * self.cols, self.path_cap * 4)     # path buffer too small: redo
                continue
            break
        return n

    def walk_bufs(self):
        """(cells [n][cap], len [n], plen [n], turns [n], status [n]) device buffers of the last walk."""
        return self._bufs[1:]

    def walk_iteration(self, iter_num, ant0=0, n=None):
        """walk_iteration_dev + host copies of the (plen[n], turns[n]) columns (turns -1 for a failed ant)."""
        self.walk_iteration_dev(iter_num, ant0, n)
        return self._bufs[3].download(), self._bufs[4].download()

    def ant_path(self, local_idx):
        dc, dl = self._bufs[1], self._bufs[2]
        L = int(dl.read(local_idx, 1)[0])
        return CellPath(dc.read(local_idx * self.path_cap, L), self.cols)

    def _construct_ant_solution_maaco(self, ant_id, current_iteration_num):
        """MAACO.py:278-302 for one ant -> (path, length, turns)."""
        plen, turns = self.walk_iteration(current_iteration_num, ant_id, 1)
        if turns[0] < 0:
            return [], INF, INF
        return self.ant_path(0).tolist(), float(plen[0]), int(turns[0])

    def update_pheromone(self, n=None):
        """MAACO.py:304-332 with the walks of the last walk_iteration call (one pass over tau)."""
        dc, dl, dp = self._bufs[1], self._bufs[2], self._bufs[3]
        n = self._bufs[0][0] if n is None else n
        self.engine.maaco_update(n, self.path_cap, dc, dl, dp, self.best_path_length_overall)

    def iterate_dev(self, iter_num, ant0=0, n=None):
        """One iteration of solve_path_planning (MAACO.py:340-359) for ants [ant0, ant0 + n), everything enqueued back to back on
        the device -- walks, best-of-iteration scan, take-over test (:351-358), one-pass pheromone update -- with ONE 104-byte
        block back (the path row of a new overall best stays in HBM until `best_path_overall` is read).  -> ib_len."""
        n = self.num_ants if n is None else n
        while True:
            dc, dl, dp, dt, ds = self._alloc(n)
            r = self.engine.maaco_iterate(iter_num, self.seed, ant0, n, self.path_cap, dc, dl, dp, dt, ds,
                                          self.best_path_length_overall, self.best_path_turns_overall)
            if r["overflow_agents"] and self.path_cap < self.rows * self.cols:
                self.path_cap = min(self.rows * self.cols, self.path_cap * 4)     # path rows too small: redo (tau was left untouched)
                continue
            if r["overflow_agents"]:
                raise RuntimeError("pathfit: path capacity overflow in a MAACO walk")
            break
        if r["took"] and r["ib_idx"] >= 0:
            self.best_path_length_overall = r["best_len"]
            self._best_on_device = True                                    # (k_maaco_best_take copied the row on the device)
            self.best_path_turns_overall = int(r["best_turns"]) if r["best_turns"] != INF else INF
        self.convergence_curve_data.append(self.best_path_length_overall if self.best_path_length_overall != INF else None)
        return r["ib_len"]

    def solve_path_planning(self):
        for iter_num in range(1, self.num_iterations + 1):
            ib_len = self.iterate_dev(iter_num)                                          # MAACO.py:340-359
            if self.verbose and (iter_num % 10 == 0 or iter_num == 1 or iter_num == self.num_iterations):
                print(f"MAACO Iter {iter_num}/{self.num_iterations}: Iter Best L={ib_len:.2f}, "
                      f"Overall Best L={self.best_path_length_overall:.2f}, T={self.best_path_turns_overall}")
        return self.best_path_overall, self.best_path_length_overall, self.best_path_turns_overall

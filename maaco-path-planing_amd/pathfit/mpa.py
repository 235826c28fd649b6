"""MPA with the reference's constructor / solve_path_planning() surface
(MPA.py:10-18, :320-448).  The population lives in HBM as strided paths and an
iteration never leaves it: the stable sort by fitness (:333,:412) is a device
radix sort of the list order, the elite (:334) is copied into a device buffer,
the phase sweep (:339-377: target-cell proposal, two A* stitches, scoring), the
memory step (:381-384) and the FADs sweep (:387-410) are device batches.  The
host keeps the scalars: CF (:336) and the 4-level best-so-far tie-break
(:415-437) on the 5 stats of the iteration's best predator; its path is read
only when the best improves.  Device-to-host traffic per iteration: the doubt
count of the proposals (4 B), the work counters (72 B), the best row (44 B).

Per-predator streams: (seed, DOM_MPA, iter, i) for the phase sweep and
(seed, DOM_MPA_FADS, iter, i) for FADs, with i the predator's index in the
fitness-sorted population of that iteration (the reference's loop index).
"""
import math

import numpy as np

from ._lib import MpaParams
from .engine import Engine, score_params
from .env import START_NODE_VAL, TARGET_NODE_VAL, find_marker
from .paths import CellPath

INF = float("inf")


def levy_sigma(beta):
    """MPA.py:251-253 (Mantegna)."""
    num = math.gamma(1 + beta) * math.sin(math.pi * beta / 2)
    den = math.gamma((1 + beta) / 2) * beta * (2 ** ((beta - 1) / 2))
    return (num / den) ** (1 / beta) if den > 1e-9 else 1.0


class MPA:
    def __init__(self, grid, num_predators, num_iterations, FADs_rate=0.2, P_const=0.5, levy_beta=1.5,
                 turn_penalty_factor=0.1, safety_penalty_factor=0.05, min_safe_distance=1.5, allow_diagonal_moves=True,
                 restrict_diagonal_near_obstacle=True, diagonal_obstacle_penalty=1000.0, engine=None, device=0, seed=0,
                 verbose=False, agent0=0, n_local=None, fused=True):
        self.grid = np.array(grid, dtype=int)
        self.rows, self.cols = self.grid.shape
        self.num_predators, self.num_iterations = num_predators, num_iterations
        self.FADs_rate, self.P_const, self.levy_beta = FADs_rate, P_const, levy_beta
        self.turn_penalty_factor_mpa = turn_penalty_factor
        self.safety_penalty_factor_mpa = safety_penalty_factor
        self.min_safe_distance_mpa = min_safe_distance
        self.allow_diagonal_moves = allow_diagonal_moves
        self.restrict_diagonal_near_obstacle = restrict_diagonal_near_obstacle
        self.diagonal_obstacle_penalty_val = diagonal_obstacle_penalty
        self.start_node = find_marker(self.grid, START_NODE_VAL, "MPA")
        self.target_node = find_marker(self.grid, TARGET_NODE_VAL, "MPA")
        self.obstacle_nodes = np.argwhere(self.grid == 1)
        self.best_path_overall = []
        self.best_path_length_overall = INF
        self.best_path_turns_overall = INF
        self.best_safety_penalty_overall = INF
        self.best_diag_penalty_overall = INF
        self.best_fitness_overall = INF
        self.convergence_curve_data = []
        self.seed, self.verbose = int(seed), verbose
        self.n_local = int(n_local) if n_local else int(num_predators)   # predators stored on this GPU (sharding)
        self.fused = bool(fused)   # one work queue for the phase sweep + FADs candidates (pf_mpa_iter_batch)
        self.engine = engine if engine is not None else Engine(self.grid, device)
        self._s = self.start_node[0] * self.cols + self.start_node[1]
        self._t = self.target_node[0] * self.cols + self.target_node[1]
        self._sp = score_params(1, restrict_diagonal_near_obstacle, turn_penalty_factor, safety_penalty_factor,
                                min_safe_distance, diagonal_obstacle_penalty)
        self.engine.mpa_setup(MpaParams(float(P_const), float(levy_beta), levy_sigma(levy_beta), float(FADs_rate),
                                        int(num_predators), self._s, self._t, int(bool(allow_diagonal_moves)),
                                        int(bool(restrict_diagonal_near_obstacle))), self._sp)
        self.path_cap = min(self.rows * self.cols, 8 * (self.rows + self.cols) + 64)
        self._init_population()

    # ------------------------------------------------------------------
    def _init_population(self):
        """MPA._initialize_population_with_safety (MPA.py:231-245): N identical A*(start, target) paths."""
        e, N = self.engine, self.n_local
        while True:
            paths, st = e.astar_host(1, [self._s], [self._t], None, path_cap=self.path_cap,
                                     allow_diag=self.allow_diagonal_moves, restrict_corner=self.restrict_diagonal_near_obstacle)
            if st[0] == 3 and self.path_cap < self.rows * self.cols:
                self.path_cap = min(self.rows * self.cols, self.path_cap * 4)
                continue
            break
        p = paths[0]
        if len(p) == 0:                                                  # :235-236
            p = np.array([self._s, self._t] if self.grid[self.target_node] != 1 else [self._s], np.int32)
        stats = e.score_host([p], self._sp)[0]
        cap = self.path_cap
        cells = np.zeros((N, cap), np.int32)
        cells[:, :len(p)] = p
        self.d_cells = e.put(cells)
        self.d_len = e.put(np.full(N, len(p), np.int32))
        self.d_stats = e.put(np.tile(stats, (N, 1)))
        self.d_cand_cells, self.d_cand_len = e.buf((N, cap), np.int32), e.buf(N, np.int32)
        self.d_cand_stats, self.d_status = e.buf((N, 5), np.float64), e.buf(N, np.int32)
        self.d_c2_cells, self.d_c2_len, self.d_c2_stats = e.buf((N, cap), np.int32), e.buf(N, np.int32), e.buf((N, 5), np.float64)
        self.d_order = e.put(np.arange(N, dtype=np.int32))               # the list: sorted position -> storage slot
        self._sorted = False                                             # (see _sort)
        self.d_gidx = e.put(np.arange(N, dtype=np.int32))                # single GPU: every predator is local
        self._el_cells, self._el_len, self._el_stats = e.mpa_elite_bufs()

    @property
    def order(self):
        """sorted position -> storage slot (host copy on demand)."""
        return self.d_order.download()

    @property
    def _stats_host(self):
        return self.d_stats.download()

    @property
    def population(self):
        """The reference's list of dicts, in the current (sorted) order; materialised on demand."""
        cells, lens, stats = self.d_cells.download(), self.d_len.download(), self.d_stats.download()
        out = []
        for slot in self.order:
            s = stats[slot]
            out.append({"path": CellPath(cells[slot, :lens[slot]].copy(), self.cols), "length": float(s[0]),
                        "turns": int(s[1]), "safety_penalty": float(s[2]), "diag_penalty": float(s[3]),
                        "fitness": float(s[4])})
        return out

    def _path_of_slot(self, slot):
        L = int(self.d_len.read(slot, 1)[0])
        return self.d_cells.read(slot * self.path_cap, L)

    def _sort(self):
        """list.sort(key=fitness) is stable (MPA.py:321,:333,:412): device sort of the list order.  The sort at the start of an
        iteration (:333) re-sorts the list the end of the previous one (:412) left sorted on the same keys -- nothing -- and is
        skipped while no sweep has touched the population since."""
        if getattr(self, "_sorted", False):
            return
        self.engine.sort_order_by_key(self.n_local, self.d_stats, 5, 4, self.d_order)
        self._sorted = True

    def _best_row(self):
        """(storage slot, stats[5]) of population[0] -- two small reads."""
        slot = int(self.d_order.read(0, 1)[0])
        return slot, self.d_stats.read(slot * 5, 5)

    def _update_best(self, s, slot):
        self.best_fitness_overall = float(s[4])
        self.best_path_overall = CellPath(self._path_of_slot(slot), self.cols).tolist()
        self.best_path_length_overall = float(s[0])
        self.best_path_turns_overall = int(s[1])
        self.best_safety_penalty_overall = float(s[2])
        self.best_diag_penalty_overall = float(s[3])

    def step(self, it):
        """One iteration of MPA.py:332-440 (it is 1-based)."""
        e, N, cap = self.engine, self.n_local, self.path_cap
        self._sort()                                                     # :333
        e.mpa_pick_elite(cap, self.d_cells, self.d_len, self.d_stats, self.d_order)   # :334 elite = population[0].copy()
        ratio = it / self.num_iterations
        CF = 0.0 if ratio >= 1.0 else ((1.0 - ratio) ** (2.0 * ratio) if ratio > 0 else 1.0)   # :336
        phase = 1 if it <= self.num_iterations / 3 else (2 if it <= 2 * self.num_iterations / 3 else 3)
        el_c, el_s = self._el_cells.ptr, self._el_stats.ptr
        self._sorted = False                                             # the sweep rewrites the population
        if self.fused:
            e.mpa_iter(phase, CF, it, self.seed, N, cap, self.d_cells, self.d_len, self.d_stats, self.d_gidx, self.d_order,
                       el_c, -1, el_s, self.d_cand_cells, self.d_cand_len, self.d_cand_stats,
                       self.d_c2_cells, self.d_c2_len, self.d_c2_stats, self.d_status)     # :339-410 in one queue
            self._check_overflow()
        else:
            elite_len = int(self._el_len.read(0, 1)[0])
            e.mpa_phase(phase, CF, it, self.seed, N, cap, self.d_cells, self.d_len, self.d_stats, self.d_gidx, self.d_order,
                        el_c, elite_len, el_s, self.d_cand_cells, self.d_cand_len, self.d_cand_stats, self.d_status)
            self._check_overflow()
            e.mpa_memory(N, cap, self.d_order, self.d_cand_cells, self.d_cand_len, self.d_cand_stats,
                         self.d_cells, self.d_len, self.d_stats)            # :381-384
            e.mpa_fads(CF, it, self.seed, N, cap, self.d_gidx, self.d_order, self.d_cells, self.d_len, self.d_stats, self.d_status)   # :387-410
            self._check_overflow()
        self._sort()                                                     # :412
        slot, s = self._best_row()
        # :415-437 best-so-far with the 4-level tie-break
        if s[4] < self.best_fitness_overall:
            self._update_best(s, slot)
        elif abs(s[4] - self.best_fitness_overall) < 1e-9:
            bl, bt, bs, bd = (self.best_path_length_overall, self.best_path_turns_overall,
                              self.best_safety_penalty_overall, self.best_diag_penalty_overall)
            if s[0] < bl:
                self._update_best(s, slot)
            elif abs(s[0] - bl) < 1e-9 and s[1] < bt:
                self._update_best(s, slot)
            elif abs(s[0] - bl) < 1e-9 and abs(s[1] - bt) < 1e-9 and s[2] < bs:
                self._update_best(s, slot)
            elif abs(s[0] - bl) < 1e-9 and abs(s[1] - bt) < 1e-9 and abs(s[2] - bs) < 1e-9 and s[3] < bd:
                self._update_best(s, slot)
        self.convergence_curve_data.append(
            self.best_fitness_overall if self.best_fitness_overall != INF else
            (self.convergence_curve_data[-1] if self.convergence_curve_data and self.convergence_curve_data[-1] is not None else None))
        return s

    def _check_overflow(self):
        n = self.engine.counters()["overflow_agents"]      # counted on the device: no status column leaves HBM
        if n:
            raise RuntimeError("pathfit: scratch/path capacity overflow on %d predators (path_cap=%d)" % (n, self.path_cap))

    def solve_path_planning(self):
        self._sort()                                                     # :321
        slot, s0 = self._best_row()
        self._update_best(s0, slot)                                      # :322-329
        self.convergence_curve_data.append(self.best_fitness_overall if self.best_fitness_overall != INF else None)
        for it in range(1, self.num_iterations + 1):
            s = self.step(it)
            if self.verbose and (it % 10 == 0 or it == 1 or it == self.num_iterations):
                print(f"MPA Iter {it}/{self.num_iterations}: IterBest Fit={s[4]:.2f}; OverallBest Fit={self.best_fitness_overall:.2f}")
        return (self.best_path_overall, self.best_path_length_overall, self.best_path_turns_overall,
                self.best_safety_penalty_overall, self.best_diag_penalty_overall, self.best_fitness_overall)

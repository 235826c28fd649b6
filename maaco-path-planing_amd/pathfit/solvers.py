"""AStarSolver / GASolver / PSOSolver with the reference's constructor and solve()
surface (astar.py:10-33, ga_solver.py:8-15,162, pso.py:8-15,163); the per-agent
decode -> A*-stitch -> score hot path runs on the GPU through libpathfit.so.

What stays on the host, as in the reference: selection / crossover / mutation
(ga_solver.py:136-160), best-of bookkeeping and the convergence lists.  Random
numbers come from per-agent keyed streams (pathfit/rng.py); PSO is evaluated
synchronously per sweep (every particle sees the sweep-start gbest).
"""
import numpy as np

from . import rng as pfrng
from .engine import Engine, score_params
from .env import START_NODE_VAL, TARGET_NODE_VAL, find_marker
from .paths import CellPath, cells_of
from . import _lib


def ga_select_native(seed, gen, fitness, tournament_size):
    """GASolver._selection (ga_solver.py:136-142) for one generation in native code: indices of the chosen parents."""
    fit = np.ascontiguousarray(fitness, np.float64)
    out = np.empty(len(fit), np.int32)
    if _lib.lib().pf_ga_select(int(seed), int(gen), len(fit), int(tournament_size), fit.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("pf_ga_select: bad arguments")
    return out


def ga_random_chromosomes_native(seed, attempt0, n, W, occ):
    """GASolver._create_chromosome for attempts attempt0..attempt0+n-1 (ga_solver.py:55-56) -> int32 cells [n][W]."""
    occ = np.ascontiguousarray(occ, np.uint8)
    out = np.empty((n, W), np.int32)
    if _lib.lib().pf_ga_random_chromosomes(int(seed), int(attempt0), int(n), int(W), occ.ctypes.data, occ.shape[0], occ.shape[1],
                                           out.ctypes.data) != 0:
        raise ValueError("pf_ga_random_chromosomes: bad arguments")
    return out


def ga_breed_native(seed, gen, crossover_rate, mutation_rate, occ, parent_cells):
    """Crossover + mutation of a whole generation (ga_solver.py:144-160, 186-194, 48-53) in native code:
    parent_cells int32 [n][W] in parent order -> child cells int32 [n][W]."""
    pc = np.ascontiguousarray(parent_cells, np.int32)
    occ = np.ascontiguousarray(occ, np.uint8)
    n, W = pc.shape
    out = np.empty((n, W), np.int32)
    if _lib.lib().pf_ga_breed(int(seed), int(gen), n, W, float(crossover_rate), float(mutation_rate), occ.ctypes.data,
                              occ.shape[0], occ.shape[1], pc.ctypes.data, out.ctypes.data) != 0:
        raise ValueError("pf_ga_breed: bad arguments")
    return out

INF = float("inf")


class BasePathfinder:
    """helper.BasePathfinder (helper.py:115-161) minus the matplotlib half."""

    def __init__(self, grid, start_node, target_node, turn_penalty_factor, safety_penalty_factor, min_safe_distance,
                 allow_diagonal_moves, restrict_diagonal_near_obstacle_policy, diagonal_obstacle_penalty_value,
                 engine=None, device=0, seed=0):
        self.grid = np.array(grid, dtype=int)
        self.rows, self.cols = self.grid.shape
        self.start_node, self.target_node = start_node, target_node
        self.obstacle_nodes = np.argwhere(self.grid == 1)
        self.turn_penalty_factor = turn_penalty_factor
        self.safety_penalty_factor = safety_penalty_factor
        self.min_safe_distance = min_safe_distance
        self.allow_diagonal_moves = allow_diagonal_moves
        self.restrict_diagonal_near_obstacle_policy = restrict_diagonal_near_obstacle_policy
        self.diagonal_obstacle_penalty_value = diagonal_obstacle_penalty_value
        self.convergence_curve = []
        self.seed = int(seed)
        self.engine = engine if engine is not None else Engine(self.grid, device)
        self._sp = score_params(0, restrict_diagonal_near_obstacle_policy, turn_penalty_factor, safety_penalty_factor,
                                min_safe_distance, diagonal_obstacle_penalty_value)

    def _cell(self, rc):
        return int(rc[0]) * self.cols + int(rc[1])

    def _calculate_stats_for_path(self, path):
        """helper.py:138 -> (path, length, turns, safety, diag, fitness)."""
        cells = cells_of(path, self.cols)
        st = self.engine.score_host([cells], self._sp)[0]
        if cells.size == 0:
            return [], INF, 0, 0.0, 0.0, INF
        return path, float(st[0]), int(st[1]), float(st[2]), float(st[3]), float(st[4])

    @staticmethod
    def _stats_tuple(path, st):
        if len(path) == 0:
            return [], INF, 0, 0.0, 0.0, INF
        return path, float(st[0]), int(st[1]), float(st[2]), float(st[3]), float(st[4])


class AStarSolver(BasePathfinder):
    def __init__(self, grid, turn_penalty_factor=0.1, safety_penalty_factor=0.05, min_safe_distance=1.5,
                 allow_diagonal_moves=True, restrict_diagonal_near_obstacle_policy=True,
                 diagonal_obstacle_penalty_value=1000.0, engine=None, device=0, seed=0):
        g = np.asarray(grid)
        start_node = find_marker(g, START_NODE_VAL, "AStar")
        target_node = find_marker(g, TARGET_NODE_VAL, "AStar")
        super().__init__(grid, start_node, target_node, turn_penalty_factor, safety_penalty_factor, min_safe_distance,
                         allow_diagonal_moves, restrict_diagonal_near_obstacle_policy, diagonal_obstacle_penalty_value,
                         engine, device, seed)
        self.astar_strictly_restricts_corners = self.restrict_diagonal_near_obstacle_policy

    def solve(self, start_node_override=None, target_node_override=None, nodes_to_avoid=None):
        """astar.py:33-101 for one query (batched form: Engine.astar_host)."""
        s = start_node_override if start_node_override else self.start_node
        t = target_node_override if target_node_override else self.target_node
        inb = lambda n: 0 <= n[0] < self.rows and 0 <= n[1] < self.cols
        if not inb(s) or not inb(t):
            return self._calculate_stats_for_path([])
        avoid = [np.array([self._cell(a) for a in nodes_to_avoid if inb(a)], np.int32)] if nodes_to_avoid else None
        paths, st = self.engine.astar_host(0, [self._cell(s)], [self._cell(t)], avoid,
                                           path_cap=self.rows * self.cols if self.rows * self.cols <= 1 << 16 else None,
                                           allow_diag=self.allow_diagonal_moves,
                                           restrict_corner=self.astar_strictly_restricts_corners)
        if st[0] == 3:
            paths, st = self.engine.astar_host(0, [self._cell(s)], [self._cell(t)], avoid, path_cap=self.rows * self.cols,
                                               allow_diag=self.allow_diagonal_moves,
                                               restrict_corner=self.astar_strictly_restricts_corners)
        path = CellPath(paths[0], self.cols).tolist()
        res = self._calculate_stats_for_path(path)
        if len(path) > 1:
            self.convergence_curve.append(res[1])      # astar.py:70: g of the goal == path length
        return res


class DijkstraSolver(BasePathfinder):
    def __init__(self, grid, turn_penalty_factor=0.1, safety_penalty_factor=0.05, min_safe_distance=1.5,
                 allow_diagonal_moves=True, restrict_diagonal_near_obstacle_policy=True,
                 diagonal_obstacle_penalty_value=1000.0, engine=None, device=0, seed=0):
        g = np.asarray(grid)
        start_node = find_marker(g, START_NODE_VAL, "Dijkstra")
        target_node = find_marker(g, TARGET_NODE_VAL, "Dijkstra")
        super().__init__(grid, start_node, target_node, turn_penalty_factor, safety_penalty_factor, min_safe_distance,
                         allow_diagonal_moves, restrict_diagonal_near_obstacle_policy, diagonal_obstacle_penalty_value,
                         engine, device, seed)
        self.dijkstra_strictly_restricts_corners = self.restrict_diagonal_near_obstacle_policy

    def solve(self, start_node_override=None, target_node_override=None, nodes_to_avoid=None):
        """dijkstra.py:32-97 for one query: AStarSolver's loop with heap entries (g, node) (batched form: Engine.astar_host, variant 2)."""
        s = start_node_override if start_node_override else self.start_node
        t = target_node_override if target_node_override else self.target_node
        inb = lambda n: 0 <= n[0] < self.rows and 0 <= n[1] < self.cols
        if not inb(s) or not inb(t):
            return self._calculate_stats_for_path([])
        avoid = [np.array([self._cell(a) for a in nodes_to_avoid if inb(a)], np.int32)] if nodes_to_avoid else None
        paths, st = self.engine.astar_host(2, [self._cell(s)], [self._cell(t)], avoid,
                                           path_cap=self.rows * self.cols if self.rows * self.cols <= 1 << 16 else None,
                                           allow_diag=self.allow_diagonal_moves,
                                           restrict_corner=self.dijkstra_strictly_restricts_corners)
        if st[0] == 3:
            paths, st = self.engine.astar_host(2, [self._cell(s)], [self._cell(t)], avoid, path_cap=self.rows * self.cols,
                                               allow_diag=self.allow_diagonal_moves,
                                               restrict_corner=self.dijkstra_strictly_restricts_corners)
        path = CellPath(paths[0], self.cols).tolist()
        res = self._calculate_stats_for_path(path)
        if len(path) > 1:
            self.convergence_curve.append(res[1])      # dijkstra.py:67: g of the goal == path length
        return res


class _WaypointSolver(BasePathfinder):
    """Shared decode + score batch for GA / PSO."""

    def _path_cap(self):
        return min(self.rows * self.cols, 16 * (self.rows + self.cols) + 64)

    def _evaluate(self, wp_cells=None, wp_pos=None):
        """-> (list[CellPath], stats ndarray[n,5], feasible mask).  Retries with the full R*C capacity if a
        path outgrows the default buffer."""
        cap = self._path_cap()
        paths, st, stats = self.engine.decode_host(self._cell(self.start_node), self._cell(self.target_node),
                                                   wp_cells=wp_cells, wp_pos=wp_pos, sp=self._sp, path_cap=cap,
                                                   allow_diag=self.allow_diagonal_moves,
                                                   restrict_corner=self.restrict_diagonal_near_obstacle_policy)
        if (st == 3).any():
            cap = self.rows * self.cols
            paths, st, stats = self.engine.decode_host(self._cell(self.start_node), self._cell(self.target_node),
                                                       wp_cells=wp_cells, wp_pos=wp_pos, sp=self._sp, path_cap=cap,
                                                       allow_diag=self.allow_diagonal_moves,
                                                       restrict_corner=self.restrict_diagonal_near_obstacle_policy)
            if (st == 3).any():
                raise RuntimeError("pathfit: open-list scratch overflow on %d agents" % int((st == 3).sum()))
        cps = [CellPath(p, self.cols) for p in paths]
        return cps, stats, np.array([len(p) > 0 for p in paths])


class GASolver(_WaypointSolver):
    def __init__(self, grid, num_generations, population_size, num_waypoints_per_chromosome, mutation_rate,
                 crossover_rate, tournament_size=3, turn_penalty_factor=0.1, safety_penalty_factor=0.05,
                 min_safe_distance=1.5, allow_diagonal_moves=True, restrict_diagonal_near_obstacle_policy=True,
                 diagonal_obstacle_penalty_value=1000.0, engine=None, device=0, seed=0, verbose=False, comm=None):
        g = np.asarray(grid)
        start_node = find_marker(g, START_NODE_VAL, "GA")
        target_node = find_marker(g, TARGET_NODE_VAL, "GA")
        super().__init__(grid, start_node, target_node, turn_penalty_factor, safety_penalty_factor, min_safe_distance,
                         allow_diagonal_moves, restrict_diagonal_near_obstacle_policy, diagonal_obstacle_penalty_value,
                         engine, device, seed)
        self.num_generations = num_generations
        self.population_size = population_size
        self.num_waypoints = num_waypoints_per_chromosome
        self.mutation_rate = mutation_rate
        self.crossover_rate = crossover_rate
        self.tournament_size = tournament_size
        # the connector obeys GA's diagonal policy (ga_solver.py:38-44); its weights are irrelevant when stitching
        self.path_connector = AStarSolver(self.grid, 0, 0, 0, allow_diagonal_moves,
                                          restrict_diagonal_near_obstacle_policy, 0, engine=self.engine)
        self._gd = None
        self.population = []
        self.best_solution_overall = {"fitness": INF, "path": []}
        self.verbose = verbose
        self._free = self.grid != 1
        self.native_operators = True      # selection / crossover / mutation in native code (device kernels / host C), not Python
        self.device_loop = True           # the whole generation in HBM (select, breed, decode, assemble, sort): SURVEY.md 8 f1/f2
        self.comm = comm                  # pathfit.dist.Comm: individuals sharded over ranks by child index (None: one GPU)

    # `population` is the reference's public list of dicts (ga_solver.py:33); while the device loop runs it lives in HBM
    # and is materialised only when somebody reads it
    @property
    def population(self):
        gd = getattr(self, "_gd", None)
        if gd is not None and gd.get("stale"):
            self._population = self._materialize_population()
            gd["stale"] = False
        return self._population

    @population.setter
    def population(self, v):
        self._population = v
        if getattr(self, "_gd", None) is not None:
            self._gd["stale"] = False

    def _decode_rows(self, chroms):
        """paths of chromosomes whose row is not stored on this rank: a path is decode(chromosome) (deterministic)."""
        if len(chroms) == 0:
            return []
        cps, _, _ = self._evaluate(wp_cells=np.ascontiguousarray(chroms, np.int32))
        return cps

    def _materialize_population(self):
        d, N, W, cap = self._gd, self.population_size, self.num_waypoints, self._gd["cap"]
        chrom = d["chrom_all"].download().reshape(N, W)
        stats = d["stats_all"].download().reshape(N, 5)
        gorder = d["gorder"].download()
        lo, hi, cur = d["lo"], d["hi"], d["cur"]
        cells = d["cells"][cur].download().reshape(-1, cap)
        lens = d["len"][cur].download()
        paths = {}
        missing = []
        for sid in gorder:
            sid = int(sid)
            if lo <= sid < hi and lens[sid - lo] >= 0:
                paths[sid] = CellPath(cells[sid - lo, :lens[sid - lo]].copy(), self.cols)
            else:
                missing.append(sid)
        for sid, cp in zip(missing, self._decode_rows(chrom[missing]) if missing else []):
            paths[sid] = cp
        C_ = self.cols
        out = []
        for sid in gorder:
            sid = int(sid)
            s = stats[sid]
            out.append({"chromosome": [(int(c) // C_, int(c) % C_) for c in chrom[sid]], "path": paths[sid], "fitness": float(s[4]),
                        "length": float(s[0]), "turns": int(s[1]), "safety_penalty": float(s[2]), "diag_penalty": float(s[3]),
                        "_cells": chrom[sid].copy()})
        return out

    def _solve_device(self):
        """ga_solver.py:178-213 with the population in HBM: selection, crossover + mutation, decode + stitch + score,
        the child-or-parent assembly and the stable sort are device work; per generation the host reads back the work
        counters, the head of the sorted list (4 B) and its 5 stats (40 B).  Sharded: individuals are owned by child
        index; C3 = all_gather of the new chromosomes and stats (N x (4 W + 40) B)."""
        from .dist import shard_range
        e, N, W, c = self.engine, self.population_size, self.num_waypoints, self.comm
        world, rank = (c.world, c.rank) if c is not None else (1, 0)
        if c is not None and c.transport == "rccl" and c.engine is None:
            c.attach(e)
        lo, hi = shard_range(N, rank, world)
        n = hi - lo
        m = max(n, 1)
        cap = self._path_cap()
        pop = self._population
        chrom = np.stack([self._chrom_cells(x) for x in pop]).astype(np.int32)
        stats = np.array([[x["length"], x["turns"], x["safety_penalty"], x["diag_penalty"], x["fitness"]] for x in pop], np.float64)
        d = {"cap": cap, "lo": lo, "hi": hi, "cur": 0, "stale": False}
        d["chrom_all"], d["stats_all"], d["fit_all"] = e.put(chrom), e.put(stats), e.put(stats[:, 4].copy())
        d["iota"] = e.put(np.arange(N, dtype=np.int32))
        d["gorder"], d["psid"] = e.put(np.arange(N, dtype=np.int32)), e.buf(N, np.int32)
        d["kid_chrom"], d["kid_cells"] = e.buf((m, W), np.int32), e.buf((m, cap), np.int32)
        d["kid_len"], d["kid_st"], d["kid_stats"] = e.buf(m, np.int32), e.buf(m, np.int32), e.buf((m, 5), np.float64)
        d["cells"] = [e.buf((m, cap), np.int32), e.buf((m, cap), np.int32)]
        d["len"] = [e.buf(m, np.int32), e.buf(m, np.int32)]
        d["chrom_loc"], d["stats_loc"] = e.buf((m, W), np.int32), e.buf((m, 5), np.float64)
        cur_cells = np.zeros((m, cap), np.int32); cur_len = np.zeros(m, np.int32)
        for i in range(n):
            cc = cells_of(pop[lo + i]["path"], self.cols)
            if len(cc) > cap:
                raise RuntimeError("pathfit: path capacity overflow in GA initialisation")
            cur_cells[i, :len(cc)] = cc; cur_len[i] = len(cc)
        d["cells"][0].upload(cur_cells); d["len"][0].upload(cur_len)
        self._gd = d
        s_cell, t_cell = self._cell(self.start_node), self._cell(self.target_node)
        counts = [shard_range(N, r, world)[1] - shard_range(N, r, world)[0] for r in range(world)]
        best_fit = self.best_solution_overall["fitness"]
        best = None                                   # (storage id, chromosome cells, stats) of an improvement made in the loop
        for gen in range(self.num_generations):
            cur = d["cur"]
            e.ga_select(self.seed, gen, N, self.tournament_size, d["fit_all"], d["gorder"], d["psid"])            # :181
            e.ga_breed(self.seed, gen, N, W, self.crossover_rate, self.mutation_rate, d["chrom_all"], d["psid"], lo, n, d["kid_chrom"])   # :186-194
            if n:                                                                                               # :198-200 the hot path
                e.decode_batch(n, W, s_cell, t_cell, cap, d["kid_cells"], d["kid_len"], d["kid_st"], d["kid_chrom"], None, self._sp,
                               d["kid_stats"], self.allow_diagonal_moves, self.restrict_diagonal_near_obstacle_policy)
                if e.counters()["overflow_agents"]:
                    raise RuntimeError("pathfit: scratch/path capacity overflow in GA decode")
            e.ga_assemble(n, W, cap, lo, d["kid_len"], d["kid_chrom"], d["kid_stats"], d["kid_cells"], d["psid"], d["chrom_all"],
                          d["stats_all"], d["cells"][cur], d["len"][cur], lo, hi, d["chrom_loc"], d["stats_loc"], d["cells"][1 - cur],
                          d["len"][1 - cur])                                                                    # :201-205
            d["cur"] = 1 - cur
            if c is not None:
                c.all_gather(d["chrom_loc"], 0, d["chrom_all"], counts, W)                                      # C3
                c.all_gather(d["stats_loc"], 0, d["stats_all"], counts, 5)
            else:
                d["chrom_all"].copy_from(0, d["chrom_loc"], 0, N * W)
                d["stats_all"].copy_from(0, d["stats_loc"], 0, N * 5)
            e.gather_col(N, d["stats_all"], 5, 4, d["fit_all"])
            d["gorder"].copy_from(0, d["iota"], 0, N)
            e.sort_order_by_key(N, d["fit_all"], 1, 0, d["gorder"])                                             # :209 stable sort
            d["stale"] = True
            gid = int(d["gorder"].read(0, 1)[0])
            s5 = d["stats_all"].read(gid * 5, 5)
            if s5[4] < best_fit:                                                                                # :212-213
                best_fit = float(s5[4])
                best = (gid, d["chrom_all"].read(gid * W, W), s5)
                own = lo <= gid < hi
                L = int(d["len"][d["cur"]].read(gid - lo, 1)[0]) if own else -1
                self._best_row = d["cells"][d["cur"]].read((gid - lo) * cap, L) if L >= 0 else None
            self.convergence_curve.append(best_fit)
            if self.verbose and ((gen + 1) % 10 == 0 or gen == 0 or gen == self.num_generations - 1):
                print(f"GA Gen {gen + 1}/{self.num_generations}: BestFit={best_fit:.2f}")
        if best is not None:
            gid, ch, s5 = best
            row = self._best_row if self._best_row is not None else self._decode_rows(ch[None, :])[0].cells
            C_ = self.cols
            self.best_solution_overall = {"chromosome": [(int(x) // C_, int(x) % C_) for x in ch], "path": CellPath(row, self.cols),
                                          "fitness": float(s5[4]), "length": float(s5[0]), "turns": int(s5[1]),
                                          "safety_penalty": float(s5[2]), "diag_penalty": float(s5[3]), "_cells": ch}
        res = self.best_solution_overall
        path = res["path"].tolist() if isinstance(res["path"], CellPath) else res["path"]
        return (path, res["length"], res["turns"], res["safety_penalty"], res["diag_penalty"], res["fitness"])

    # ---- host-side genetic operators (ga_solver.py:48-56,136-160), per-agent streams ----
    def _generate_random_waypoint(self, r):
        while True:
            rr = r.randint(0, self.rows - 1)
            cc = r.randint(0, self.cols - 1)
            if self._free[rr, cc]:
                return (rr, cc)

    def _create_chromosome(self, r):
        return [self._generate_random_waypoint(r) for _ in range(self.num_waypoints)]

    def _reconstruct_path_from_chromosome(self, chromosome):
        """ga_solver.py:58-93 for one chromosome."""
        if not chromosome:
            return self.path_connector.solve(self.start_node, self.target_node)[0]
        wp = np.array([[self._cell(w) for w in chromosome]], np.int32)
        return self._evaluate(wp_cells=wp)[0][0].tolist()

    def _individuals(self, chroms, cps, stats):
        return [{"chromosome": c, "path": p, "fitness": float(s[4]), "length": float(s[0]), "turns": int(s[1]),
                 "safety_penalty": float(s[2]), "diag_penalty": float(s[3])} for c, p, s in zip(chroms, cps, stats)]

    def _chrom_cells(self, ind):
        """int32 cells of an individual's chromosome (cached on the individual)."""
        cc = ind.get("_cells")
        if cc is None:
            cc = ind["_cells"] = np.array([self._cell(w) for w in ind["chromosome"]], np.int32)
        return cc

    def _initialize_population(self):
        """ga_solver.py:95-133; attempt k draws its chromosome from stream (seed, DOM_INIT, 0, k)."""
        self.population = []
        max_total_attempts = self.population_size * 20
        k = 0
        while len(self.population) < self.population_size and k < max_total_attempts:
            batch = min(max_total_attempts - k, max(self.population_size - len(self.population), 32) * 2)
            if self.native_operators and self.num_waypoints > 0:
                wp = ga_random_chromosomes_native(self.seed, k, batch, self.num_waypoints, self.grid == 1)
                C_ = self.cols
                chroms = [[(c // C_, c % C_) for c in row] for row in wp.tolist()]
            else:
                chroms = [self._create_chromosome(pfrng.AgentRandom(self.seed, pfrng.DOM_INIT, 0, k + i)) for i in range(batch)]
                wp = np.array([[self._cell(w) for w in c] for c in chroms], np.int32).reshape(batch, self.num_waypoints)
            cps, stats, feas = self._evaluate(wp_cells=wp)
            for j, (ind, ok) in enumerate(zip(self._individuals(chroms, cps, stats), feas)):
                if ok and len(self.population) < self.population_size:
                    ind["_cells"] = wp[j]
                    self.population.append(ind)
            k += batch
        if not self.population and self.num_waypoints > 0:
            path_direct = self._reconstruct_path_from_chromosome([])
            if path_direct and path_direct[0] == self.start_node and path_direct[-1] == self.target_node:
                _, l, t, sp, dp, f = self._calculate_stats_for_path(path_direct)
                self.population.append({"chromosome": [], "path": path_direct, "fitness": f, "length": l, "turns": t,
                                        "safety_penalty": sp, "diag_penalty": dp})
        if not self.population:
            self.population = [{"chromosome": [], "path": [], "fitness": INF, "length": INF, "turns": 0,
                                "safety_penalty": 0, "diag_penalty": 0}] * self.population_size
            return False
        r = pfrng.AgentRandom(self.seed, pfrng.DOM_INIT, 1, 0)
        while len(self.population) < self.population_size:
            self.population.append(r.choice(self.population).copy())
        self.population.sort(key=lambda x: x["fitness"])
        return True

    def _selection(self, gen):
        """ga_solver.py:136-142; the whole generation's tournaments draw from stream (seed, DOM_GA_SELECT, gen, 0)."""
        out = []
        r = pfrng.AgentRandom(self.seed, pfrng.DOM_GA_SELECT, gen, 0)
        for _ in range(self.population_size):
            tournament = r.sample(self.population, min(self.tournament_size, len(self.population)))
            out.append(min(tournament, key=lambda x: x["fitness"]))
        return out

    def _crossover(self, p1, p2, r):
        if r.random() < self.crossover_rate and self.num_waypoints > 0:
            point = r.randint(1, self.num_waypoints - 1) if self.num_waypoints > 1 else 0
            if point > 0:
                return p1[:point] + p2[point:], p2[:point] + p1[point:]
        return list(p1), list(p2)

    def _mutate(self, chromosome, r):
        if not chromosome:
            return []
        m = list(chromosome)
        for i in range(len(m)):
            if r.random() < self.mutation_rate:
                m[i] = self._generate_random_waypoint(r)
        return m

    def solve(self):
        if self.num_waypoints == 0:
            path = self._reconstruct_path_from_chromosome([])
            stats = self._calculate_stats_for_path(path)
            self.best_solution_overall = {"path": stats[0], "fitness": stats[5], "length": stats[1], "turns": stats[2],
                                          "safety_penalty": stats[3], "diag_penalty": stats[4]}
            self.convergence_curve.append(stats[5])
            return stats
        if not self._initialize_population():
            return [], INF, 0, 0.0, 0.0, INF
        self.best_solution_overall = self.population[0].copy()
        self.convergence_curve.append(self.best_solution_overall["fitness"])
        N = self.population_size
        Wn = self.num_waypoints
        if self.native_operators and self.device_loop and len(self.population) == N and \
                all(len(x["chromosome"]) == Wn for x in self.population) and hasattr(self.engine, "ga_select"):
            return self._solve_device()
        for gen in range(self.num_generations):
            kid_cells = None
            if self.native_operators and len(self.population) == N and all(len(x["chromosome"]) == Wn for x in self.population):
                # the genetic operators in native code (same streams, same CPython derivations: tests/test_ga_native.py)
                pidx = ga_select_native(self.seed, gen, [x["fitness"] for x in self.population], self.tournament_size)
                parents = [self.population[i] for i in pidx]
                kid_cells = ga_breed_native(self.seed, gen, self.crossover_rate, self.mutation_rate, self.grid == 1,
                                            np.stack([self._chrom_cells(x) for x in parents]))
                C_ = self.cols
                kids = [[(int(c) // C_, int(c) % C_) for c in row] for row in kid_cells.tolist()]
                owners = [(parents[(i & ~1) % N], parents[((i & ~1) + 1) % N]) for i in range(N)]
            else:
                parents = self._selection(gen)
                # children of pair j come from stream (seed, DOM_GA, gen, j) (ga_solver.py:186-194)
                kids, owners = [], []
                idx = pair = 0
                while len(kids) < N:
                    p1, p2 = parents[idx % len(parents)], parents[(idx + 1) % len(parents)]
                    idx += 2
                    r = pfrng.AgentRandom(self.seed, pfrng.DOM_GA, gen, pair)
                    pair += 1
                    c1, c2 = self._crossover(p1["chromosome"], p2["chromosome"], r)
                    for c in (self._mutate(c1, r), self._mutate(c2, r)):
                        if len(kids) < N:
                            kids.append(c)
                            owners.append((p1, p2))
            # ---- the hot path: decode + stitch + score every child on the GPU ----
            with_wp = [i for i, c in enumerate(kids) if c]
            cps, stats, feas = [None] * N, np.zeros((N, 5)), np.zeros(N, bool)
            if with_wp:
                wp = kid_cells if kid_cells is not None else np.array([[self._cell(w) for w in kids[i]] for i in with_wp], np.int32)
                a, b, c_ = self._evaluate(wp_cells=wp)
                for j, i in enumerate(with_wp):
                    cps[i], stats[i], feas[i] = a[j], b[j], c_[j]
            new_pop = []
            for i in range(N):
                if kids[i] == [] :
                    path = self._reconstruct_path_from_chromosome([])
                    ok = bool(path) and path[0] == self.start_node and path[-1] == self.target_node
                    st = self._calculate_stats_for_path(path) if ok else None
                    if ok:
                        new_pop.append({"chromosome": [], "path": path, "fitness": st[5], "length": st[1], "turns": st[2],
                                        "safety_penalty": st[3], "diag_penalty": st[4]})
                        continue
                elif feas[i]:
                    new_pop.append(self._individuals([kids[i]], [cps[i]], [stats[i]])[0])
                    if kid_cells is not None:
                        new_pop[-1]["_cells"] = kid_cells[i]
                    continue
                p1, p2 = owners[i]                                   # ga_solver.py:204-205
                new_pop.append(p1 if len(new_pop) % 2 == 0 else p2)
            self.population = new_pop
            self.population.sort(key=lambda x: x["fitness"])
            if self.population[0]["fitness"] < self.best_solution_overall["fitness"]:
                self.best_solution_overall = self.population[0].copy()
            self.convergence_curve.append(self.best_solution_overall["fitness"])
            if self.verbose and ((gen + 1) % 10 == 0 or gen == 0 or gen == self.num_generations - 1):
                b = self.best_solution_overall
                print(f"GA Gen {gen + 1}/{self.num_generations}: BestFit={b['fitness']:.2f} (L:{b['length']:.1f}, T:{b['turns']})")
        res = self.best_solution_overall
        path = res["path"].tolist() if isinstance(res["path"], CellPath) else res["path"]
        return (path, res["length"], res["turns"], res["safety_penalty"], res["diag_penalty"], res["fitness"])


class PSOSolver(_WaypointSolver):
    def __init__(self, grid, num_iterations, num_particles, num_waypoints_per_particle, w, c1, c2,
                 turn_penalty_factor=0.1, safety_penalty_factor=0.05, min_safe_distance=1.5, allow_diagonal_moves=True,
                 restrict_diagonal_near_obstacle_policy=True, diagonal_obstacle_penalty_value=1000.0, engine=None,
                 device=0, seed=0, verbose=False, asynchronous=True, comm=None):
        g = np.asarray(grid)
        start_node = find_marker(g, START_NODE_VAL, "PSO")
        target_node = find_marker(g, TARGET_NODE_VAL, "PSO")
        super().__init__(grid, start_node, target_node, turn_penalty_factor, safety_penalty_factor, min_safe_distance,
                         allow_diagonal_moves, restrict_diagonal_near_obstacle_policy, diagonal_obstacle_penalty_value,
                         engine, device, seed)
        self.num_iterations = num_iterations
        self.num_particles = num_particles
        self.num_waypoints = num_waypoints_per_particle
        self.w, self.c1, self.c2 = w, c1, c2
        self.max_vel = max(1.0, 0.15 * max(self.rows, self.cols))      # pso.py:34
        self.path_connector = AStarSolver(self.grid, 0, 0, 0, allow_diagonal_moves,
                                          restrict_diagonal_near_obstacle_policy, 0, engine=self.engine)
        self._d, self._gbest_dev, self._particles_stale = None, None, False
        self.comm = comm                  # pathfit.dist.Comm: particles sharded over ranks in contiguous blocks (None: one GPU)
        self._lo, self._hi = 0, num_particles
        self.particles = []
        self.gbest_particle_data = {"fitness": INF, "path": [], "position": []}
        self.verbose = verbose
        # asynchronous=True reproduces pso.py:222-229 exactly (a particle sees the gbest updated by the particles
        # before it in the same sweep) by speculate-and-repair; False = one batch per sweep (sweep-start gbest)
        self.asynchronous = bool(asynchronous)
        # asynchronous mode speculates on ALL particles not yet final in one batch; a bound (particles per round) changes how
        # much is speculated, never the result (tests/test_gpu_fullsize.py checks exactly that at 2048 particles)
        self.max_speculation = None

    def _reconstruct_path_from_position(self, position_waypoints_float):
        """pso.py:56-94 for one particle."""
        if not len(position_waypoints_float):
            return self.path_connector.solve(self.start_node, self.target_node)[0]
        wp = np.asarray(position_waypoints_float, np.float64).reshape(1, -1, 2)
        return self._evaluate(wp_pos=wp)[0][0].tolist()

    def _initialize_particles(self):
        """pso.py:97-161: attempt k draws position then velocity from stream (seed, DOM_INIT, 0, k)."""
        W, N = self.num_waypoints, self.num_particles
        pos, vel, cps, stats = [], [], [], []
        k, max_total = 0, N * 20
        while len(pos) < N and k < max_total:
            batch = min(max_total - k, max(N - len(pos), 32) * 2)
            P = np.zeros((batch, W, 2)); V = np.zeros((batch, W, 2))
            for i in range(batch):
                r = pfrng.AgentRandom(self.seed, pfrng.DOM_INIT, 0, k + i)
                P[i] = [[r.uniform(0, self.rows - 1), r.uniform(0, self.cols - 1)] for _ in range(W)]     # :50-51
                V[i] = [[r.uniform(-self.max_vel / 5, self.max_vel / 5) for _ in range(2)] for _ in range(W)]   # :105
            a, b, feas = self._evaluate(wp_pos=P)
            for i in range(batch):
                if feas[i] and len(pos) < N:
                    pos.append(P[i]); vel.append(V[i]); cps.append(a[i]); stats.append(b[i])
            k += batch
        if not pos and W > 0:                                            # :126-143 fallback: the direct A* path as one particle
            direct = self._reconstruct_path_from_position([])
            if direct and direct[0] == self.start_node and direct[-1] == self.target_node:
                cells = cells_of(direct, self.cols)
                st = self.engine.score_host([cells], self._sp)[0]
                pos.append(np.zeros((W, 2))); vel.append(np.zeros((W, 2)))
                cps.append(CellPath(cells, self.cols)); stats.append(st)
        if not pos:
            return False
        r = pfrng.AgentRandom(self.seed, pfrng.DOM_INIT, 1, 0)
        while len(pos) < N:                                              # :159-160 random copies
            j = r.randrange(len(pos))
            pos.append(pos[j].copy()); vel.append(vel[j].copy()); cps.append(cps[j]); stats.append(stats[j])
        self._pos, self._vel = np.array(pos), np.array(vel)
        self._pbest, self._pbest_fit = self._pos.copy(), np.array([s[4] for s in stats])
        self._pbest_path, self._pbest_stats = list(cps), [np.array(s) for s in stats]
        self._cur_path, self._cur_stats = list(cps), [np.array(s) for s in stats]
        g = int(np.argmin(self._pbest_fit))                              # first minimum == the sequential :121 scan
        self._set_gbest(g, self._pos[g], cps[g], stats[g])
        self._sync_particles()
        return True

    def _set_gbest(self, idx, position, path, st):
        self._gbest = {"fitness": float(st[4]), "path": path, "position": [list(p) for p in position],
                       "length": float(st[0]), "turns": int(st[1]), "safety_penalty": float(st[2]),
                       "diag_penalty": float(st[3])}
        self._gbest_dev = None

    # gbest_particle_data / particles are the reference's public attributes (pso.py:37-38); while a solve is running
    # they live in HBM and are materialised only when somebody reads them
    def fetch_gbest(self):
        """Sharded runs: bring the gbest's stats and path row from the rank that owns them to every rank.  A COLLECTIVE --
        every rank must call it (finish() does); reading `gbest_particle_data` never communicates."""
        d = getattr(self, "_gbest_dev", None)
        c = self.comm
        # (every term of this test is the same on all ranks: _gowner is the TRUE owner everywhere, and the flag is set and
        # cleared by collective steps only -- so either every rank enters the broadcasts below, with the same root, or none does)
        if d is None or c is None or c.world == 1 or self._gowner < 0 or self._gbest_everywhere:
            return
        c.broadcast(self._d["gstats"], 0, 5, self._gowner)              # device row to device row (the owner's k_pso_commit wrote it)
        c.broadcast(self._d["gpath"], 0, 1, self._gowner)               # the length, then exactly that many cells
        L = int(self._d["gpath"].read(0, 1)[0])
        c.broadcast(self._d["gpath"], 1, L, self._gowner)
        self._gbest_dev = dict(d, host=False)
        self._gbest_everywhere = True                                    # every rank holds the row now (until a sweep moves the gbest)

    @property
    def gbest_particle_data(self):
        if getattr(self, "_gbest_dev", None) is not None:
            d = self._gbest_dev
            e, c = self.engine, self.comm
            pos = e.read(self._d["gb"].ptr, self.num_waypoints * 2, np.float64).reshape(-1, 2)   # (every rank holds the position)
            if c is not None and c.world > 1 and c.rank != self._gowner and not self._gbest_everywhere:
                # purely local view: the path row and the stats live on the owner until fetch_gbest() (a collective) is called
                return {"fitness": float(d["fitness"]), "position": pos.tolist(), "path": None, "index": d["idx"], "owner_rank": self._gowner}
            if not d.get("host"):
                L = int(self._d["gpath"].read(0, 1)[0])                  # the row k_pso_commit left in HBM: length, cells, five stats
                cells = self._d["gpath"].read(1, L)
                self._set_gbest(d["idx"], pos, CellPath(cells, self.cols), self._d["gstats"].read(0, 5))
                # the device-side record stays: whether fetch_gbest() has anything to do must not depend on which ranks happened to
                # READ the property (a monitoring rank would otherwise leave the collective the others still enter)
                self._gbest_dev = dict(d, host=True)
        return self._gbest

    @gbest_particle_data.setter
    def gbest_particle_data(self, v):
        self._gbest, self._gbest_dev = v, None

    @property
    def particles(self):
        if getattr(self, "_d", None) is not None and self._particles_stale:
            self._download_state()
            self._sync_particles()
            self._particles_stale = False
        return self._particles

    @particles.setter
    def particles(self, v):
        self._particles = v

    def _sync_particles(self):
        self._particles = [{"position": self._pos[i].tolist(), "velocity": self._vel[i].tolist(),
                            "pbest_position": self._pbest[i].tolist(), "pbest_fitness": float(self._pbest_fit[i]),
                            "pbest_path": self._pbest_path[i], "current_path": self._cur_path[i],
                            "current_fitness": float(self._cur_stats[i][4])} for i in range(len(self._pos))]

    def _download_state(self):
        """HBM -> the host mirrors behind `particles` (bulk copies; only on demand and at the end of solve()).  A sharded
        run mirrors this rank's particles [lo, hi)."""
        d, n, cap = self._d, self._hi - self._lo, self._cap
        self._pos, self._vel = d["pos"].download(), d["vel"].download()
        self._pbest, self._pbest_fit = d["pb"].download(), d["pbf"].download()
        cells, lens, stats = d["cells"].download(), d["len"].download(), d["stats"].download()
        pbc, pbl = d["pb_cells"].download(), d["pb_len"].download()
        self._cur_path = [CellPath(cells[i, :lens[i]].copy(), self.cols) for i in range(n)]
        self._cur_stats = [stats[i] for i in range(n)]
        self._pbest_path = [CellPath(pbc[i, :pbl[i]].copy(), self.cols) for i in range(n)]

    def begin(self):
        """Initialise the swarm (pso.py:97-161) and move it into HBM; False if no particle could be built.  Sharded
        (comm.world > 1): every rank replays the same keyed initialisation and keeps its block [lo, hi) of the particles."""
        from .dist import shard_range
        self._d = None
        self._it = 0
        c = self.comm
        self._lo, self._hi = shard_range(self.num_particles, c.rank, c.world) if c is not None else (0, self.num_particles)
        if not self._initialize_particles():
            return False
        self.convergence_curve.append(self._gbest["fitness"])
        e, W = self.engine, self.num_waypoints
        if c is not None and c.transport == "rccl" and c.engine is None:
            c.attach(e)
        lo, hi = self._lo, self._hi
        n = hi - lo
        self._pos, self._vel, self._pbest, self._pbest_fit = self._pos[lo:hi], self._vel[lo:hi], self._pbest[lo:hi], self._pbest_fit[lo:hi]
        self._cur_path, self._cur_stats, self._pbest_path = self._cur_path[lo:hi], self._cur_stats[lo:hi], self._pbest_path[lo:hi]
        self._sync_particles()
        cap = self._cap = self._path_cap()
        m = max(n, 1)
        d = {}
        d["pos"], d["vel"], d["pb"] = e.buf((m, W, 2), np.float64), e.buf((m, W, 2), np.float64), e.buf((m, W, 2), np.float64)
        d["pbf"] = e.buf(m, np.float64)
        d["gb"] = e.put(np.array(self._gbest["position"], np.float64))
        d["cells"], d["len"], d["st"] = e.buf((m, cap), np.int32), e.buf(m, np.int32), e.buf(m, np.int32)
        d["stats"], d["imp"] = e.buf((m, 5), np.float64), e.buf(m, np.int32)
        d["pos0"], d["vel0"] = e.buf((m, W, 2), np.float64), e.buf((m, W, 2), np.float64)
        d["gpath"] = e.buf(cap + 1, np.int32)
        d["gstats"] = e.buf(5, np.float64)
        d["pb_cells"], d["pb_len"] = e.buf((m, cap), np.int32), e.buf(m, np.int32)
        if n:
            d["pos"].upload(self._pos); d["vel"].upload(self._vel); d["pb"].upload(self._pbest); d["pbf"].upload(self._pbest_fit)
            # current / pbest paths start as the initial paths (pso.py:111-117)
            cur = np.zeros((n, cap), np.int32); ln = np.zeros(n, np.int32)
            for i, cp in enumerate(self._cur_path):
                cc = cells_of(cp, self.cols)
                if len(cc) > cap:
                    raise RuntimeError("pathfit: path capacity overflow in PSO initialisation")
                cur[i, :len(cc)] = cc; ln[i] = len(cc)
            d["cells"].upload(cur); d["len"].upload(ln)
            d["stats"].upload(np.array(self._cur_stats, np.float64).reshape(n, 5))
            d["pb_cells"].upload(cur); d["pb_len"].upload(ln)
        self._d = d
        self._gowner = -1                     # rank whose d["gpath"] holds the gbest path (-1: the host copy from the initialisation)
        self._gbest_everywhere = False        # fetch_gbest() has brought the owner's row to every rank and no sweep has moved the gbest since
        self._particles_stale = False
        return True

    def sweep(self):
        """One iteration of pso.py:178-231 over the whole swarm, resident in HBM: update -> decode/stitch -> score ->
        pbest -> gbest.  Speculate that no particle from `cur` on improves gbest: evaluate them in one batch with the
        current gbest.  Everything up to and including the first improver p* is exact; gbest moves to p* and the
        particles after it are rolled back and re-evaluated (their draws are keyed per particle, so the re-evaluation
        consumes the same random numbers).  Synchronous mode commits the whole batch.  Sharded, the first improver is
        the smallest GLOBAL index over the ranks (16 B per rank per round), its owner broadcasts the new gbest position.
        Device-to-host traffic: 16 bytes per round (the improver scan) + the decode batch's counter block; no path, position or
        stats column leaves HBM, and a round is four launches (update, decode, scan, commit) with no copy in between."""
        e, N, W, d, cap, it, c = self.engine, self.num_particles, self.num_waypoints, self._d, self._cap, self._it, self.comm
        world = c.world if c is not None else 1
        rank = c.rank if c is not None else 0
        lo, hi = self._lo, self._hi
        s_cell, t_cell = self._cell(self.start_node), self._cell(self.target_node)
        st_sz = W * 2 * 8
        gfit = self._gbest_dev["fitness"] if self._gbest_dev is not None else self._gbest["fitness"]
        cur = 0                                                            # global index of the first particle not yet final
        while cur < N:
            a0 = max(lo, cur)                                              # my particles [a0, hi) are evaluated this round
            top = hi if not (self.asynchronous and self.max_speculation) else min(hi, cur + int(self.max_speculation))
            m = max(top - a0, 0)
            idx, fit = -1, INF
            if m:
                l0 = a0 - lo
                # update (the pre-update position / velocity stay in pos0 / vel0 for the roll-back) -> decode + score -> scan
                e.pso_update_keep_raw(m, W, self.w, self.c1, self.c2, self.max_vel, d["pos"].at(l0 * W * 2), d["vel"].at(l0 * W * 2),
                                      d["pb"].at(l0 * W * 2), d["gb"].ptr, self.seed, it, a0, d["pos0"].at(l0 * W * 2), d["vel0"].at(l0 * W * 2))
                e.decode_raw(m, W, s_cell, t_cell, cap, d["cells"].at(l0 * cap), d["len"].at(l0), d["st"].at(l0), d["pos"].at(l0 * W * 2),
                             self._sp, d["stats"].at(l0 * 5), self.allow_diagonal_moves, self.restrict_diagonal_near_obstacle_policy)
                idx, fit, ovf = e.pso_scan(m, d["stats"].at(l0 * 5), d["len"].at(l0), d["st"].at(l0), d["pbf"].at(l0), gfit,
                                           0 if self.asynchronous else 1)
                if ovf:
                    raise RuntimeError("pathfit: scratch/path capacity overflow in PSO decode")
            mine = a0 + idx if idx >= 0 else -1                            # global index of my improver
            if world > 1:
                allv = c.all_gather_host([mine if mine >= 0 else INF, fit])
                if self.asynchronous:
                    r_star = int(np.argmin(allv[:, 0]))                    # the smallest global index = the first improver
                else:                                                      # smallest fitness, then smallest index (rank order = index order)
                    r_star = int(np.lexsort((allv[:, 0], allv[:, 1]))[0])
                p_star = int(allv[r_star, 0]) if np.isfinite(allv[r_star, 0]) else -1
                f_star = float(allv[r_star, 1])
            else:
                r_star, p_star, f_star = rank, mine, fit
            last_eval = N - 1 if not (self.asynchronous and self.max_speculation) else min(N, cur + int(self.max_speculation)) - 1
            upto = (p_star if p_star >= 0 else last_eval) if self.asynchronous else N - 1   # last particle whose evaluation is final
            if m:
                # ONE launch commits the round (k_pso_commit): pso.py:216-220 for my final particles (pbest position, fitness, path
                # row), pso.py:222-229 for the improver if it is mine (position, stats and path row stay in HBM), roll-back of the
                # particles that were evaluated on a gbest that has moved.  No copy, nothing synchronises.
                l0 = a0 - lo
                k = min(max(min(upto, hi - 1) - a0 + 1, 0), m)             # my evaluated particles that are final now
                j = p_star - a0 if (p_star >= 0 and r_star == rank) else -1
                e.pso_commit_raw(m, W, cap, k, j, d["pos"].at(l0 * W * 2), d["vel"].at(l0 * W * 2), d["pos0"].at(l0 * W * 2),
                                 d["vel0"].at(l0 * W * 2), d["stats"].at(l0 * 5), d["len"].at(l0), d["cells"].at(l0 * cap),
                                 d["pb"].at(l0 * W * 2), d["pbf"].at(l0), d["pb_cells"].at(l0 * cap), d["pb_len"].at(l0),
                                 d["gb"].ptr, d["gstats"].ptr, d["gpath"].ptr)
            if p_star >= 0:                                                # pso.py:222-229: gbest moves to particle p*
                self._gbest_dev = {"idx": p_star, "fitness": f_star, "host": False}
                if world > 1:
                    c.broadcast(d["gb"], 0, W * 2, r_star)                 # the new gbest position: W x 16 B
                self._gowner = r_star
                self._gbest_everywhere = False
                gfit = f_star
            cur = upto + 1
        self._particles_stale = True
        self._it += 1
        self.convergence_curve.append(gfit)
        return gfit

    def solve(self):
        if self.num_waypoints == 0:
            path = self._reconstruct_path_from_position([])
            stats = self._calculate_stats_for_path(path)
            self.gbest_particle_data = {"path": stats[0], "fitness": stats[5], "length": stats[1], "turns": stats[2],
                                        "safety_penalty": stats[3], "diag_penalty": stats[4], "position": []}
            self.convergence_curve.append(stats[5])
            return stats
        if not self.begin():
            return [], INF, 0, 0.0, 0.0, INF
        for it in range(self.num_iterations):
            g = self.sweep()
            if self.verbose and ((it + 1) % 10 == 0 or it == 0 or it == self.num_iterations - 1):
                print(f"PSO Iter {it + 1}/{self.num_iterations}: GBestFit={g:.2f}")
        return self.finish()

    def finish(self):
        self._download_state()
        self._sync_particles()
        self._particles_stale = False
        self.fetch_gbest()
        res = self.gbest_particle_data
        path = res["path"].tolist() if isinstance(res["path"], CellPath) else res["path"]
        res["path"] = path
        return (path, res.get("length", INF), res.get("turns", INF), res.get("safety_penalty", INF),
                res.get("diag_penalty", INF), res["fitness"])

"""Population sharding over GPUs (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" for the CPU tests).

The agents of one iteration are independent, so the data path has NO
collective: every rank walks / rebuilds / decodes its contiguous block of
agents on its own GPU with streams keyed by the GLOBAL agent index (results do
not depend on the partition).  One small exchange per iteration remains
(SURVEY.md 8e):
  C2  best-of-iteration: all_gather of per-agent (length, turns) or fitness
      (N x 16 B), then the sequential best scan every rank repeats, then a
      broadcast of the winner's path from its owner.
  C1  MAACO pheromone: the reference deposits ant by ant (MAACO.py:306-311), so
      the exact result is a FOLD in global ant order, not a sum: rank k adds its
      ants' deposits onto the matrix it receives from rank k-1 (send/recv ring,
      R*C*8 B per hop), the last rank clips and broadcasts.  `strict=False`
      replaces the fold by all_reduce(SUM) of per-rank deltas (ulp-level
      deviation from the reference's summation order).
  MPA  all_gather of the fitness column to rebuild the global stable sort, and
      a broadcast of the elite path (MPA.py:333-334).
"""
import numpy as np

INF = float("inf")


def shard_range(n_total, rank, world):
    """Contiguous block [a0, a1) of agents for `rank`."""
    base, rem = divmod(n_total, world)
    a0 = rank * base + min(rank, rem)
    return a0, a0 + base + (1 if rank < rem else 0)


class Comm:
    """Thin wrapper so the same code runs single-process (no torch needed) and multi-process."""

    def __init__(self, dist=None, device=None):
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1

    def _t(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device) if self.device is not None else t

    def all_gather_concat(self, arr, counts):
        """Concatenate per-rank 1-D/2-D host arrays (row counts known to every rank)."""
        if self.world == 1:
            return np.asarray(arr)
        import torch
        arr = np.ascontiguousarray(arr)
        mx = max(counts)
        pad = np.zeros((mx,) + arr.shape[1:], arr.dtype)
        pad[: arr.shape[0]] = arr
        mine = self._t(pad)
        outs = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)

    def broadcast(self, arr, src):
        if self.world == 1:
            return np.asarray(arr)
        t = self._t(arr)
        self.dist.broadcast(t, src)
        return t.cpu().numpy()

    def broadcast_obj_shape(self, n, src):
        return int(self.broadcast(np.array([n], np.int64), src)[0])

    def send(self, arr, dst):
        self.dist.send(self._t(arr), dst)

    def recv(self, like, src):
        t = self._t(np.empty_like(like))
        self.dist.recv(t, src)
        return t.cpu().numpy()

    def all_reduce_sum(self, arr):
        if self.world == 1:
            return np.asarray(arr)
        t = self._t(arr)
        self.dist.all_reduce(t)
        return t.cpu().numpy()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()


def owner_of(gidx, counts):
    """(rank, local index) of global agent gidx under contiguous sharding."""
    acc = 0
    for r, c in enumerate(counts):
        if gidx < acc + c:
            return r, gidx - acc
        acc += c
    raise IndexError(gidx)


def maaco_best_scan_host(plen, turns, best_len=INF, best_turns=INF, best_idx=-1):
    """MAACO.py:343-349 over host arrays (turns < 0 == failed ant); pure-python twin of pf_maaco_best_scan."""
    for i in range(len(plen)):
        L = float(plen[i])
        T = INF if turns[i] < 0 else float(turns[i])
        if L < best_len:
            best_len, best_idx, best_turns = L, i, T
        elif abs(L - best_len) < 1e-9 and T < best_turns:
            best_idx, best_turns = i, T
    return best_len, best_turns, best_idx


def global_stable_order(fitness_all):
    """list.sort(key=fitness) of the concatenated population: position -> global storage id."""
    return np.argsort(np.asarray(fitness_all), kind="stable")


class ShardedMAACO:
    """MAACO.solve_path_planning (MAACO.py:334-371) with the ants of every iteration sharded over ranks."""

    def __init__(self, comm, make_local, num_ants, strict=True):
        self.comm = comm
        self.counts = [shard_range(num_ants, r, comm.world)[1] - shard_range(num_ants, r, comm.world)[0]
                       for r in range(comm.world)]
        self.a0, self.a1 = shard_range(num_ants, comm.rank, comm.world)
        self.local = make_local()            # a pathfit.MAACO on this rank's GPU (full num_ants in its params)
        self.strict = strict
        self.num_ants = num_ants

    def step(self, iter_num):
        m, c = self.local, self.comm
        n = self.a1 - self.a0
        plen, turns = m.walk_iteration(iter_num, self.a0, n)
        allp = c.all_gather_concat(plen, self.counts)
        allt = c.all_gather_concat(turns, self.counts)
        scan = getattr(m.engine, "maaco_best_scan", None)     # C twin of maaco_best_scan_host (same sequential rule)
        ib_len, ib_turns, ib_idx = scan(allp, allt, 0, INF, INF, -1) if scan else maaco_best_scan_host(allp, allt)
        take = ib_len < m.best_path_length_overall or \
            (abs(ib_len - m.best_path_length_overall) < 1e-9 and ib_turns < m.best_path_turns_overall)
        if take and ib_idx >= 0:
            r, li = owner_of(ib_idx, self.counts)
            cells = m.ant_path(li).cells if r == c.rank else np.zeros(0, np.int32)
            L = c.broadcast_obj_shape(len(cells), r)
            buf = np.zeros(L, np.int32)
            buf[: len(cells)] = cells
            cells = c.broadcast(buf, r)
            from .paths import CellPath
            if ib_len < m.best_path_length_overall:
                m.best_path_length_overall = ib_len
            m.best_path_overall = CellPath(cells, m.cols).tolist()
            m.best_path_turns_overall = int(ib_turns) if ib_turns != INF else INF
        self._update_pheromone(n)
        m.convergence_curve_data.append(m.best_path_length_overall if m.best_path_length_overall != INF else None)
        return ib_len

    def _update_pheromone(self, n):
        m, c, e = self.local, self.comm, self.local.engine
        dc, dl, dp = m._bufs[1], m._bufs[2], m._bufs[3]
        e.maaco_evaporate()                                           # every rank evaporates the same matrix
        if c.world == 1:
            e.maaco_deposit(n, m.path_cap, dc, dl, dp)
        elif self.strict:
            # ordered fold: rank r continues from rank r-1's matrix (global ant order)
            if c.rank > 0:
                e.maaco_set_pheromone(c.recv(np.empty((m.rows, m.cols)), c.rank - 1))
            e.maaco_deposit(n, m.path_cap, dc, dl, dp)
            tau = e.maaco_get_pheromone()
            if c.rank < c.world - 1:
                c.send(tau, c.rank + 1)
            tau = c.broadcast(tau, c.world - 1)
            e.maaco_set_pheromone(tau)
        else:
            base = e.maaco_get_pheromone()
            e.maaco_deposit(n, m.path_cap, dc, dl, dp)
            delta = e.maaco_get_pheromone() - base
            e.maaco_set_pheromone(base + c.all_reduce_sum(delta))
        e.maaco_clip(m.best_path_length_overall)

    def solve_path_planning(self):
        m = self.local
        for it in range(1, m.num_iterations + 1):
            self.step(it)
        return m.best_path_overall, m.best_path_length_overall, m.best_path_turns_overall


class ShardedMPA:
    """MPA.solve_path_planning (MPA.py:320-448) with predators sharded over ranks.  Every rank stores its
    block of predators; the global fitness-sorted order is rebuilt from an all_gather of the fitness column."""

    def __init__(self, comm, make_local, num_predators_total):
        self.comm = comm
        self.N = num_predators_total
        self.counts = [shard_range(self.N, r, comm.world)[1] - shard_range(self.N, r, comm.world)[0]
                       for r in range(comm.world)]
        self.off = np.concatenate([[0], np.cumsum(self.counts)])
        self.local = make_local(self.counts[comm.rank])   # pathfit.MPA with n_local predators, num_predators=N for the Levy split
        # global list order (position -> global storage id); starts as the identity like the reference's list
        self.gorder = np.arange(self.N)
        self._fit_all = None
        self._sorted = False              # gorder reflects the current stats (set by the sort that ends a step)

    def _resort(self):
        if self._sorted:                  # MPA.py:333 right after :412 of the previous iteration: nothing changed
            return
        m, c = self.local, self.comm
        fit_local = m._stats_host[:, 4]
        self._fit_all = c.all_gather_concat(fit_local, self.counts)
        self.gorder = self.gorder[np.argsort(self._fit_all[self.gorder], kind="stable")]
        self._sorted = True

    def _local_view(self):
        """gidx / slot arrays of the predators stored on this rank, in global-position order."""
        r = self.comm.rank
        lo, hi = self.off[r], self.off[r + 1]
        pos = np.flatnonzero((self.gorder >= lo) & (self.gorder < hi))
        return pos.astype(np.int32), (self.gorder[pos] - lo).astype(np.int32)

    def step(self, it):
        m, c, e = self.local, self.comm, self.local.engine
        cap = m.path_cap
        self._resort()                                                # MPA.py:333
        elite_gid = int(self.gorder[0])                               # :334
        er, eslot = owner_of(elite_gid, self.counts)
        if er == c.rank:
            ecells = m._path_of_slot(eslot); estats = m._stats_host[eslot].copy()
        else:
            ecells = np.zeros(0, np.int32); estats = np.zeros(5)
        L = c.broadcast_obj_shape(len(ecells), er)
        buf = np.zeros(max(L, 1), np.int32); buf[: len(ecells)] = ecells
        ecells = c.broadcast(buf, er)[:L]
        estats = c.broadcast(estats, er)
        d_el = e.put(np.concatenate([ecells, np.zeros(1, np.int32)]))
        d_es = e.put(estats)
        gidx, slot = self._local_view()
        n = len(gidx)
        d_gidx, d_slot = e.put(gidx if n else np.zeros(1, np.int32)), e.put(slot if n else np.zeros(1, np.int32))
        ratio = it / m.num_iterations
        CF = 0.0 if ratio >= 1.0 else ((1.0 - ratio) ** (2.0 * ratio) if ratio > 0 else 1.0)
        phase = 1 if it <= m.num_iterations / 3 else (2 if it <= 2 * m.num_iterations / 3 else 3)
        e.mpa_iter(phase, CF, it, m.seed, n, cap, m.d_cells, m.d_len, m.d_stats, d_gidx, d_slot, d_el.ptr, L, d_es.ptr,
                   m.d_cand_cells, m.d_cand_len, m.d_cand_stats, m.d_c2_cells, m.d_c2_len, m.d_c2_stats, m.d_status)
        m._check_overflow()
        m._stats_host = m.d_stats.download()
        self._sorted = False
        self._resort()                                                # :412
        return self._fit_all[self.gorder[0]]

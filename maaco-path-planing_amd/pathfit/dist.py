"""Population sharding over GPUs: one process per GPU, contiguous blocks of agents, streams keyed by the GLOBAL agent
index (results do not depend on the partition).

The agents of one iteration are independent, so the data path has NO collective; what remains is one small exchange per
iteration, on DEVICE buffers:

  C2  best-of-iteration.  all_gather of the per-agent (length, turns) / fitness columns (N x 12 B or N x 8 B), the scan
      every rank repeats on the device (24 B come back to the host), a broadcast of the winner's path row from its owner
      only when the overall best improves.
  C1  MAACO pheromone.  The reference deposits ant by ant (MAACO.py:306-311), so the exact result is a FOLD in global
      ant order, not a sum: rank k adds its ants' deposits to the matrix it receives from rank k-1.  The fold is
      PIPELINED over row chunks of tau: rank k folds chunk j while rank k+1 folds chunk j-1, so with c chunks the
      critical path is (world + c - 1) / c deposits instead of `world`; the finished chunks are broadcast from the last
      rank as they complete.  `strict=False` replaces the fold by all_reduce(SUM) of per-rank deltas (ulp-level
      deviation from the reference's summation order).
  MPA all_gather of the fitness column, the global stable sort on the device, a broadcast of the elite row
      (MPA.py:333-334).  The sweep is bounded by its longest search, so MPA scales WEAKLY only: more
      predators per step at the same step time, never a shorter step for a fixed population.
  PSO the asynchronous gbest (pso.py:222-229) couples all particles in global order: every repair round all_gathers
      (first local improver, its fitness) -- 16 B per rank --, the owner of the global first improver broadcasts its
      position (W x 16 B), the particles after it are re-evaluated.
  GA  C3: all_gather of the fitness column and the chromosomes (N x (8 + 4 W) B); selection, which samples the whole
      population (ga_solver.py:139), is then replayed identically on every rank; children are bred and decoded by the
      rank that owns their index.

Transports.  `rccl`: pf_comm_* of libpathfit.so, i.e. RCCL over xGMI called directly on the engine's stream with device
pointers -- no torch tensor, no host staging, nothing synchronises (torch.distributed, when present, only ships the
128-byte unique id).  `gloo` / `torch`: torch.distributed with host staging (CPU tensors for the gloo backend -- the
world-size-2 tests and rehearsals on one GPU --, this rank's GPU for the nccl backend -- the fallback bench.py takes if
the direct binding cannot be initialised).  All move the same bytes between the same buffers, so the solver code below
is transport independent.
"""
import os
import time

import numpy as np

INF = float("inf")


def shard_range(n_total, rank, world):
    """Contiguous block [a0, a1) of agents for `rank`."""
    base, rem = divmod(n_total, world)
    a0 = rank * base + min(rank, rem)
    return a0, a0 + base + (1 if rank < rem else 0)


def owner_of(gidx, counts):
    """(rank, local index) of global agent gidx under contiguous sharding."""
    acc = 0
    for r, c in enumerate(counts):
        if gidx < acc + c:
            return r, gidx - acc
        acc += c
    raise IndexError(gidx)


def maaco_best_scan_host(plen, turns, best_len=INF, best_turns=INF, best_idx=-1):
    """MAACO.py:343-349 over host arrays (turns < 0 == failed ant); pure-python twin of the device scan."""
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


class HostBuf:
    """numpy-backed stand-in for engine.DevBuf (same read / write / copy_from protocol): lets the exchange logic run
    against CPU fakes in the tests."""

    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype).reshape(-1)
        self.dtype = self.a.dtype
        self.ptr = None

    def at(self, off):
        raise RuntimeError("HostBuf has no device address (use the gloo transport)")

    def read(self, off, count):
        return self.a[int(off):int(off) + int(count)].copy()

    def write(self, off, arr):
        arr = np.asarray(arr, self.dtype).reshape(-1)
        self.a[int(off):int(off) + arr.size] = arr
        return self

    def copy_from(self, off, src, src_off, count):
        self.a[int(off):int(off) + int(count)] = src.read(src_off, count)
        return self

    def download(self):
        return self.a.copy()

    def upload(self, arr):
        return self.write(0, arr)


class Comm:
    """rank / world and the four collectives the solvers use, on buffer objects (DevBuf or HostBuf) + element ranges."""

    def __init__(self, dist=None, device=None, engine=None, transport=None, loopback=False):
        self.loopback = bool(loopback)   # world == 1 only: still issue every collective (a one-rank RCCL communicator) -- lets a
                                         # single GPU exercise the rccl transport's pointers, sizes and stream ordering
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist is not None else int(os.environ.get("PF_COMM_RANK", "0")) if transport == "rccl" else 0
        self.world = dist.get_world_size() if dist is not None else int(os.environ.get("PF_COMM_WORLD", "1")) if transport == "rccl" else 1
        self.engine = None
        self.transport = transport or ("gloo" if dist is not None else None)
        self.bytes_moved = 0                      # per-rank bytes handed to the transport (accounting for DESIGN.md 6)
        self.calls = 0
        self.timed = False                        # bench.py: time every collective (HIP-event spans on the engine's stream for the
        self.host_s = 0.0                         # rccl transport -- nothing synchronises --, the host clock for the host-staged ones)
        if engine is not None and self.transport == "rccl":
            self.attach(engine)

    # ---- RCCL bootstrap: rank 0 makes the 128-byte id, it reaches the others through torch.distributed or a file ----
    def attach(self, engine):
        import ctypes as C
        self.engine = engine
        if self.transport != "rccl":
            return self
        idb = (C.c_char * 128)()
        if self.rank == 0:
            if engine.L.pf_comm_unique_id(idb) != 0:
                raise RuntimeError("pf_comm_unique_id failed: " + engine.L.pf_last_error(None).decode())
        if self.dist is not None and self.world > 1:
            import torch
            t = torch.frombuffer(bytearray(bytes(idb)), dtype=torch.uint8).clone()
            if self.device is not None:
                t = t.to(self.device)
            self.dist.broadcast(t, 0)
            raw = bytes(t.cpu().numpy().tobytes())
            idb = (C.c_char * 128).from_buffer_copy(raw)
        elif self.world > 1:
            path = os.environ["PF_COMM_ID_FILE"]
            if self.rank == 0:
                with open(path + ".tmp", "wb") as f:
                    f.write(bytes(idb))
                os.replace(path + ".tmp", path)
            else:
                for _ in range(6000):
                    if os.path.exists(path):
                        break
                    time.sleep(0.01)
                idb = (C.c_char * 128).from_buffer_copy(open(path, "rb").read())
        engine._ck(engine.L.pf_comm_init(engine.h, self.rank, self.world, idb))
        return self

    def attach_checked(self, engine, timeout=120.0):
        """attach() plus one all_gather of the rank numbers through the new communicator, run in a helper thread so that a
        bootstrap that never returns cannot take the caller with it; every rank then learns, through torch.distributed,
        whether ALL ranks made it.  Returns True when the rccl transport is usable everywhere.

        Otherwise there are two cases.  (a) Every helper thread came back (librccl missing, an error code): this rank's
        transport falls back to the host-staged torch collectives -- and so does every other rank's (same verdict).
        (b) Some rank's helper is still inside the library at the deadline (`attach_stuck`, the same on every rank): that
        thread shares the engine handle and its stream with the caller, so the handle must not be used again -- the caller
        has to abandon this process for GPU work (bench.py starts a fresh one with PF_BENCH_TRANSPORT=torch).  A helper that
        wakes up after the deadline sees `cancelled`, destroys the communicator it may have made and touches neither the
        engine nor its stream."""
        import threading
        ok = [False]
        err = [None]
        cancelled = threading.Event()

        def work():
            try:
                self.attach(engine)
                if cancelled.is_set():                 # too late: the verdict is out.  No probe collective, no engine calls.
                    engine.L.pf_comm_destroy(engine.h)
                    return
                if self.world > 1:
                    mine = engine.put(np.array([self.rank], np.int32), np.int32)
                    allr = engine.buf(self.world, np.int32)
                    engine._ck(engine.L.pf_comm_all_gather(engine.h, mine.ptr, allr.ptr, 4))
                    got = allr.download()
                    if not np.array_equal(got, np.arange(self.world, dtype=np.int32)):
                        raise RuntimeError(f"pf_comm_all_gather returned {got.tolist()}")
                ok[0] = True
            except Exception as ex:            # noqa: BLE001 -- any failure means "use the other transport"
                err[0] = ex
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout)
        stuck = th.is_alive()
        if stuck:
            cancelled.set()
        good = ok[0] and not stuck
        self.attach_error = err[0] if err[0] is not None else ("timed out" if stuck else None)
        if self.dist is not None and self.world > 1:
            import torch
            t = torch.tensor([1 if good else 0, 0 if stuck else 1], dtype=torch.int32)
            if self.device is not None:
                t = t.to(self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            t = t.cpu()
            good, stuck = bool(int(t[0])), not bool(int(t[1]))
        self.attach_stuck = stuck              # on ANY rank: a peer that did initialise would wait for it in the first collective
        if not good:
            self.transport = "torch"
            self.engine = engine
        return good

    def _acct(self, nbytes):
        self.bytes_moved += int(nbytes)
        self.calls += 1

    def _t(self, arr):
        """host array -> the tensor the torch.distributed backend wants (CPU for gloo, this rank's GPU for nccl)"""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device) if self.device is not None else t

    @staticmethod
    def _n(t):
        return t.cpu().numpy()

    # ---- collectives ----
    def _all_gather_impl(self, src, src_off, dst, counts, row_elems=1):
        """dst[offsets[r] * row_elems ...] = rank r's src[src_off ... + counts[r] * row_elems): the concatenation in
        rank order of per-rank blocks of counts[r] rows."""
        mine = counts[self.rank] * row_elems
        offs = np.concatenate([[0], np.cumsum(counts)]) * row_elems
        if self.world == 1 and not (self.loopback and self.transport == "rccl"):
            dst.copy_from(0, src, src_off, mine)
            return
        isz = np.dtype(dst.dtype).itemsize
        if self.transport == "rccl":
            e = self.engine
            if len(set(counts)) == 1:
                e._ck(e.L.pf_comm_all_gather(e.h, src.at(src_off), dst.at(0), mine * isz))
                self._acct(mine * isz * (self.world - 1))
            else:                                  # uneven blocks: one broadcast per rank, straight into place
                dst.copy_from(int(offs[self.rank]), src, src_off, mine)
                for r in range(self.world):
                    nb = int(counts[r] * row_elems * isz)
                    e._ck(e.L.pf_comm_broadcast(e.h, dst.at(int(offs[r])), nb, r))
                    self._acct(nb)
            return
        import torch
        mx = max(counts) * row_elems
        pad = np.zeros(mx, dst.dtype)
        pad[:mine] = src.read(src_off, mine)
        t = self._t(pad)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        self._acct(mine * isz * (self.world - 1))
        for r in range(self.world):
            dst.write(int(offs[r]), self._n(outs[r])[: counts[r] * row_elems])

    def _broadcast_impl(self, buf, off, count, root):
        if (self.world == 1 and not (self.loopback and self.transport == "rccl")) or count == 0:
            return
        isz = np.dtype(buf.dtype).itemsize
        self._acct(count * isz)
        if self.transport == "rccl":
            e = self.engine
            e._ck(e.L.pf_comm_broadcast(e.h, buf.at(off), int(count) * isz, int(root)))
            return
        t = self._t(buf.read(off, count) if self.rank == root else np.zeros(int(count), buf.dtype))
        self.dist.broadcast(t, root)
        if self.rank != root:
            buf.write(off, self._n(t))

    def _send_impl(self, buf, off, count, peer):
        isz = np.dtype(buf.dtype).itemsize
        self._acct(count * isz)
        if self.transport == "rccl":
            e = self.engine
            e._ck(e.L.pf_comm_send(e.h, buf.at(off), int(count) * isz, int(peer)))
            return
        self.dist.send(self._t(buf.read(off, count)), peer)

    def _recv_impl(self, buf, off, count, peer):
        isz = np.dtype(buf.dtype).itemsize
        if self.transport == "rccl":
            e = self.engine
            e._ck(e.L.pf_comm_recv(e.h, buf.at(off), int(count) * isz, int(peer)))
            return
        t = self._t(np.zeros(int(count), buf.dtype))
        self.dist.recv(t, peer)
        buf.write(off, self._n(t))

    def _all_reduce_sum_f64_impl(self, buf, off, count):
        if self.world == 1 and not (self.loopback and self.transport == "rccl"):
            return
        self._acct(2 * count * 8)
        if self.transport == "rccl":
            e = self.engine
            e._ck(e.L.pf_comm_all_reduce_f64(e.h, buf.at(off), int(count), 0))
            return
        t = self._t(buf.read(off, count))
        self.dist.all_reduce(t)
        buf.write(off, self._n(t))

    def _all_gather_host_impl(self, arr):
        """A few host scalars per rank (e.g. PSO's per-round (first improver, fitness)) -> [world][k] float64."""
        a = np.ascontiguousarray(arr, np.float64).reshape(-1)
        if self.world == 1 and not (self.loopback and self.transport == "rccl"):
            return a[None, :]
        if self.transport == "rccl":
            e = self.engine
            if getattr(self, "_hs", None) is None or self._hs[0].shape[0] < a.size:
                self._hs = (e.buf(max(a.size, 8), np.float64), e.buf(max(a.size, 8) * self.world, np.float64))
            self._hs[0].write(0, a)
            e._ck(e.L.pf_comm_all_gather(e.h, self._hs[0].ptr, self._hs[1].ptr, a.size * 8))
            self._acct(a.size * 8 * (self.world - 1))
            return self._hs[1].read(0, a.size * self.world).reshape(self.world, a.size)
        import torch
        t = self._t(a)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        self._acct(a.size * 8 * (self.world - 1))
        return np.stack([self._n(o) for o in outs])

    def barrier(self):
        if self.world > 1 and self.dist is not None:
            self.dist.barrier()

    # ---- the public collectives: the implementations above, timed on request ----
    def _timed(self, fn, *a, **k):
        if not self.timed or self.world == 1:
            return fn(*a, **k)
        if self.transport == "rccl" and self.engine is not None:
            self.engine.span_begin()
            try:
                return fn(*a, **k)
            finally:
                self.engine.span_end()
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            self.host_s += time.perf_counter() - t0

    def all_gather(self, *a, **k):
        return self._timed(self._all_gather_impl, *a, **k)

    def broadcast(self, *a, **k):
        return self._timed(self._broadcast_impl, *a, **k)

    def send(self, *a, **k):
        return self._timed(self._send_impl, *a, **k)

    def recv(self, *a, **k):
        return self._timed(self._recv_impl, *a, **k)

    def all_reduce_sum_f64(self, *a, **k):
        return self._timed(self._all_reduce_sum_f64_impl, *a, **k)

    def all_gather_host(self, *a, **k):
        return self._timed(self._all_gather_host_impl, *a, **k)

    def exchange_ms(self, reset=True):
        """Milliseconds spent in the collectives since the last reset (timed = True): event time on the engine's stream for the
        rccl transport, host wall time for the host-staged ones."""
        ms = self.host_s * 1e3
        if self.transport == "rccl" and self.engine is not None and self.world > 1:
            ms += self.engine.span_total(reset)[0]
        if reset:
            self.host_s = 0.0
        return ms


def _counts(n, world):
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def _mkbuf(engine, shape, dtype):
    return engine.buf(shape, dtype)


# ======================================================================================================================
class ShardedMAACO:
    """MAACO.solve_path_planning (MAACO.py:334-371) with the ants of every iteration sharded over ranks."""

    def __init__(self, comm, make_local, num_ants, strict=True, chunks=8):
        self.comm = comm
        self.counts = _counts(num_ants, comm.world)
        self.a0, self.a1 = shard_range(num_ants, comm.rank, comm.world)
        self.local = make_local()            # a pathfit.MAACO on this rank's GPU (full num_ants in its params)
        self.strict = strict
        self.num_ants = num_ants
        self.chunks = max(1, int(chunks))
        e = self.local.engine
        if comm.transport == "rccl" and comm.engine is None:
            comm.attach(e)
        self._allp, self._allt = _mkbuf(e, num_ants, np.float64), _mkbuf(e, num_ants, np.int32)
        self._base = None

    def step(self, iter_num):
        m, c, e = self.local, self.comm, self.local.engine
        n = self.a1 - self.a0
        if c.world == 1 and not c.loopback and hasattr(m, "iterate_dev"):
            return m.iterate_dev(iter_num, self.a0, n)                  # one GPU: the whole iteration in one enqueue (no exchange)
        m.walk_iteration_dev(iter_num, self.a0, n)
        dc, dl, dp, dt, ds = m.walk_bufs()
        c.all_gather(dp, 0, self._allp, self.counts)                     # C2: 12 B per ant
        c.all_gather(dt, 0, self._allt, self.counts)
        ib_len, ib_turns, ib_idx = e.maaco_best_dev(self.num_ants, self._allp, self._allt)   # MAACO.py:343-349, same on every rank
        take = ib_len < m.best_path_length_overall or \
            (abs(ib_len - m.best_path_length_overall) < 1e-9 and ib_turns < m.best_path_turns_overall)
        if take and ib_idx >= 0:
            r, li = owner_of(ib_idx, self.counts)
            # the winner's path row travels only now, when the overall best changes: its length first, then exactly that many
            # cells.  (Never `path_cap` cells: a rank whose ant overflowed has grown ITS path_cap in walk_iteration_dev, so
            # the ranks need not agree on it -- the count of a collective must not depend on it.)
            if getattr(self, "_hdr", None) is None:
                self._hdr = _mkbuf(e, 1, np.int32)
            if r == c.rank:
                self._hdr.copy_from(0, dl, li, 1)
            c.broadcast(self._hdr, 0, 1, r)
            L = int(self._hdr.read(0, 1)[0])
            if getattr(self, "_row", None) is None or self._row_cap < L:
                self._row_cap = max(L, m.path_cap)
                self._row = _mkbuf(e, self._row_cap, np.int32)
            if r == c.rank:
                self._row.copy_from(0, dc, li * m.path_cap, L)
            c.broadcast(self._row, 0, L, r)
            from .paths import CellPath
            if ib_len < m.best_path_length_overall:
                m.best_path_length_overall = ib_len
            m.best_path_overall = CellPath(self._row.read(0, L), m.cols).tolist()
            m.best_path_turns_overall = int(ib_turns) if ib_turns != INF else INF
        self._update_pheromone(n)
        m.convergence_curve_data.append(m.best_path_length_overall if m.best_path_length_overall != INF else None)
        return ib_len

    def _update_pheromone(self, n):
        m, c, e = self.local, self.comm, self.local.engine
        dc, dl, dp = m.walk_bufs()[:3]
        tau = e.tau_buf
        RC = m.rows * m.cols
        if c.world == 1 and hasattr(e, "maaco_update"):
            e.maaco_update(n, m.path_cap, dc, dl, dp, m.best_path_length_overall)   # evaporate + ordered deposits + clip in one pass
            return
        if c.world == 1:
            e.maaco_evaporate()
            e.maaco_deposit_begin(n, m.path_cap, dc, dl, dp)
            e.maaco_deposit_cells(0, RC)
        elif self.strict:
            # ordered fold, pipelined over row chunks: rank r continues chunk j from rank r-1's matrix (global ant order)
            if c.rank == 0:
                e.maaco_evaporate()                                   # MAACO.py:305 once; the others receive the evaporated matrix
            e.maaco_deposit_begin(n, m.path_cap, dc, dl, dp)
            rows = m.rows
            k = min(self.chunks, rows)
            bounds = [(rows * j // k) * m.cols for j in range(k + 1)]
            for j in range(k):
                c0, c1 = bounds[j], bounds[j + 1]
                if c.rank > 0:
                    c.recv(tau, c0, c1 - c0, c.rank - 1)
                e.maaco_deposit_cells(c0, c1)
                if c.rank < c.world - 1:
                    c.send(tau, c0, c1 - c0, c.rank + 1)
            for j in range(k):                                        # the last rank holds the result: broadcast chunk by chunk
                c.broadcast(tau, bounds[j], bounds[j + 1] - bounds[j], c.world - 1)
        else:
            e.maaco_evaporate()
            if self._base is None:
                self._base, self._delta = _mkbuf(e, RC, np.float64), _mkbuf(e, RC, np.float64)
            self._base.copy_from(0, tau, 0, RC)
            e.maaco_deposit_begin(n, m.path_cap, dc, dl, dp)
            e.maaco_deposit_cells(0, RC)
            if hasattr(e, "vec_add"):                                  # delta and sum on device columns: nothing crosses PCIe
                e.vec_add(RC, tau, self._base, -1.0, self._delta)
                c.all_reduce_sum_f64(self._delta, 0, RC)
                e.vec_add(RC, self._base, self._delta, 1.0, tau)
            else:                                                      # (CPU fakes of the gloo tests)
                self._delta.write(0, tau.read(0, RC) - self._base.read(0, RC))
                c.all_reduce_sum_f64(self._delta, 0, RC)
                tau.write(0, self._base.read(0, RC) + self._delta.read(0, RC))
        e.maaco_clip(m.best_path_length_overall)

    def solve_path_planning(self):
        m = self.local
        for it in range(1, m.num_iterations + 1):
            self.step(it)
        return m.best_path_overall, m.best_path_length_overall, m.best_path_turns_overall


# ======================================================================================================================
class ShardedMPA:
    """MPA.solve_path_planning (MPA.py:320-448) with predators sharded over ranks.  Every rank stores its block of
    predators (global ids [lo, hi)); the global fitness-sorted list order lives on every rank's device and is rebuilt by
    an all_gather of the fitness column + the same stable device sort everywhere."""

    def __init__(self, comm, make_local, num_predators_total):
        self.comm = comm
        self.N = num_predators_total
        self.counts = _counts(self.N, comm.world)
        self.off = np.concatenate([[0], np.cumsum(self.counts)])
        self.local = make_local(self.counts[comm.rank])   # pathfit.MPA with n_local predators, num_predators=N for the Levy split
        m, e = self.local, self.local.engine
        if comm.transport == "rccl" and comm.engine is None:
            comm.attach(e)
        n = self.counts[comm.rank]
        self.d_gorder = e.put(np.arange(self.N, dtype=np.int32))      # the global list: position -> global id
        self.d_fit_all = e.buf(self.N, np.float64)
        self.d_fit_loc = e.buf(max(n, 1), np.float64)
        self.d_gidx, self.d_slot = e.buf(max(n, 1), np.int32), e.buf(max(n, 1), np.int32)
        self.d_hdr = e.buf(8, np.float64)
        self._fresh = False        # the global list is sorted on the population as it stands (set by _resort, cleared by a sweep)
        self._head = None          # ... and its head (global id, fitness) has been read

    @property
    def gorder(self):
        return self.d_gorder.download()

    def _resort(self):
        """The reference sorts at the end of an iteration (MPA.py:412) and again at the start of the next (:333) -- a stable sort of
        a list that is already sorted on the same keys, i.e. nothing: the second one (gather, all_gather, three kernels, a 16-byte
        read and its host round trip) is skipped while no sweep has touched the population since."""
        if self._fresh:
            return
        m, c, e = self.local, self.comm, self.local.engine
        n = self.counts[c.rank]
        e.gather_col(n, m.d_stats, 5, 4, self.d_fit_loc)
        c.all_gather(self.d_fit_loc, 0, self.d_fit_all, self.counts)  # 8 B per predator
        e.sort_order_by_key(self.N, self.d_fit_all, 1, 0, self.d_gorder)
        self._fresh, self._head = True, None

    def _first(self):
        """global id at the head of the list + its fitness (one 16-byte read per sort)."""
        if self._head is not None:
            return self._head
        e = self.local.engine
        if hasattr(e, "sorted_head"):
            self._head = e.sorted_head(self.d_gorder, self.d_fit_all)
        else:
            gid = int(self.d_gorder.read(0, 1)[0])                   # (CPU fakes of the gloo tests)
            self._head = (gid, float(self.d_fit_all.read(gid, 1)[0]))
        return self._head

    def step(self, it):
        m, c, e = self.local, self.comm, self.local.engine
        cap = m.path_cap
        lo, hi = int(self.off[c.rank]), int(self.off[c.rank + 1])
        n = hi - lo
        self._resort()                                                # MPA.py:333
        elite_gid, _ = self._first()                                  # :334
        er, _ = owner_of(elite_gid, self.counts)
        el_c, el_l, el_s = m._el_cells, m._el_len, m._el_stats
        if er == c.rank:
            e.mpa_pick_elite(cap, m.d_cells, m.d_len, m.d_stats, self.d_gorder, lo)
        c.broadcast(el_l, 0, 1, er)
        c.broadcast(el_s, 0, 5, er)
        c.broadcast(el_c, 0, cap, er)                                 # the elite row: path_cap x 4 B
        e.mpa_local_view(self.N, self.d_gorder, lo, hi, self.d_gidx, self.d_slot)
        ratio = it / m.num_iterations
        CF = 0.0 if ratio >= 1.0 else ((1.0 - ratio) ** (2.0 * ratio) if ratio > 0 else 1.0)
        phase = 1 if it <= m.num_iterations / 3 else (2 if it <= 2 * m.num_iterations / 3 else 3)
        self._fresh, self._head = False, None                        # the sweep rewrites the population
        e.mpa_iter(phase, CF, it, m.seed, n, cap, m.d_cells, m.d_len, m.d_stats, self.d_gidx, self.d_slot, el_c.ptr, -1, el_s.ptr,
                   m.d_cand_cells, m.d_cand_len, m.d_cand_stats, m.d_c2_cells, m.d_c2_len, m.d_c2_stats, m.d_status)
        m._check_overflow()
        self._resort()                                                # :412
        return self._first()[1]

    # ---- the solve loop with the reference's best-so-far tracking (MPA.py:320-448) on every rank ----
    def _best_row(self):
        """stats[5] of the head of the global list, broadcast from its owner; -> (gid, owner rank, stats)."""
        c = self.comm
        gid, _ = self._first()
        r, slot = owner_of(gid, self.counts)
        if r == c.rank:
            self.d_hdr.copy_from(0, self.local.d_stats, slot * 5, 5)
        c.broadcast(self.d_hdr, 0, 5, r)
        return gid, r, slot, self.d_hdr.read(0, 5)

    def _take_best(self, s, r, slot):
        """MPA._update_best_overall on every rank: the path row comes from its owner."""
        m, c, e = self.local, self.comm, self.local.engine
        if getattr(self, "_row", None) is None:
            self._row = e.buf(m.path_cap + 1, np.int32)
        if r == c.rank:
            self._row.copy_from(0, m.d_len, slot, 1)
            self._row.copy_from(1, m.d_cells, slot * m.path_cap, m.path_cap)
        c.broadcast(self._row, 0, m.path_cap + 1, r)
        L = int(self._row.read(0, 1)[0])
        from .paths import CellPath
        m.best_fitness_overall = float(s[4])
        m.best_path_overall = CellPath(self._row.read(1, L), m.cols).tolist()
        m.best_path_length_overall, m.best_path_turns_overall = float(s[0]), int(s[1])
        m.best_safety_penalty_overall, m.best_diag_penalty_overall = float(s[2]), float(s[3])

    def solve_path_planning(self):
        m = self.local
        self._resort()                                                # :321
        _, r, slot, s = self._best_row()
        self._take_best(s, r, slot)                                   # :322-329
        m.convergence_curve_data.append(m.best_fitness_overall if m.best_fitness_overall != INF else None)
        for it in range(1, m.num_iterations + 1):
            self.step(it)
            _, r, slot, s = self._best_row()
            if s[4] < m.best_fitness_overall:                         # :415-437 with the 4-level tie-break
                self._take_best(s, r, slot)
            elif abs(s[4] - m.best_fitness_overall) < 1e-9:
                bl, bt, bs, bd = (m.best_path_length_overall, m.best_path_turns_overall, m.best_safety_penalty_overall,
                                  m.best_diag_penalty_overall)
                if s[0] < bl or (abs(s[0] - bl) < 1e-9 and s[1] < bt) or \
                   (abs(s[0] - bl) < 1e-9 and abs(s[1] - bt) < 1e-9 and s[2] < bs) or \
                   (abs(s[0] - bl) < 1e-9 and abs(s[1] - bt) < 1e-9 and abs(s[2] - bs) < 1e-9 and s[3] < bd):
                    self._take_best(s, r, slot)
            m.convergence_curve_data.append(
                m.best_fitness_overall if m.best_fitness_overall != INF else
                (m.convergence_curve_data[-1] if m.convergence_curve_data and m.convergence_curve_data[-1] is not None else None))
        return (m.best_path_overall, m.best_path_length_overall, m.best_path_turns_overall, m.best_safety_penalty_overall,
                m.best_diag_penalty_overall, m.best_fitness_overall)


# ======================================================================================================================
def ShardedPSO(comm, grid, **kw):
    """PSOSolver with the swarm sharded over the ranks of `comm` in contiguous blocks of particles; the asynchronous gbest
    (pso.py:222-229) is repaired across ranks (see PSOSolver.sweep).  solve() returns the same tuple on every rank."""
    from .solvers import PSOSolver
    return PSOSolver(grid, comm=comm, **kw)


def ShardedGA(comm, grid, **kw):
    """GASolver with the individuals sharded over the ranks of `comm` by child index; C3 (all_gather of fitness and
    chromosomes) feeds the replicated tournament selection (see GASolver._solve_device)."""
    from .solvers import GASolver
    return GASolver(grid, comm=comm, **kw)

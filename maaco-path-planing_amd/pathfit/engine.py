"""Engine: one (device, grid) handle of libpathfit.so with numpy-friendly batch calls.

Device buffers are explicit (``DevBuf``) so that populations stay resident in
HBM between calls; the ``*_host`` conveniences stage numpy arrays for the
facades and tests.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Counters, MaacoParams, MpaParams, PathfitError, ScoreParams

INF = float("inf")
ST_OK, ST_INFEASIBLE, ST_STEP_CAP, ST_OVERFLOW, ST_KEPT = 0, 1, 2, 3, 4


class DevBuf:
    """A typed device allocation owned by an Engine."""

    def __init__(self, eng, shape, dtype):
        self.eng = eng
        self.shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = C.c_void_p()
        eng._ck(eng.L.pf_dev_alloc(eng.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        a = np.ascontiguousarray(arr, self.dtype)
        assert a.nbytes <= self.nbytes, (a.nbytes, self.nbytes)
        if a.nbytes:
            self.eng._ck(self.eng.L.pf_h2d(self.eng.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, count=None):
        out = np.empty(self.shape if count is None else (count,), self.dtype)
        if out.nbytes:
            self.eng._ck(self.eng.L.pf_d2h(self.eng.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def zero(self):
        self.eng._ck(self.eng.L.pf_memset(self.eng.h, self.ptr, 0, self.nbytes))
        return self

    def at(self, elem_offset):
        return self.ptr + int(elem_offset) * self.dtype.itemsize

    def read(self, elem_offset, count):
        """Host copy of `count` elements from `elem_offset` (row reads on demand)."""
        out = np.empty(int(count), self.dtype)
        if out.nbytes:
            self.eng._ck(self.eng.L.pf_d2h(self.eng.h, out.ctypes.data, self.at(elem_offset), out.nbytes))
        return out

    def write(self, elem_offset, arr):
        a = np.ascontiguousarray(arr, self.dtype)
        if a.nbytes:
            self.eng._ck(self.eng.L.pf_h2d(self.eng.h, self.at(elem_offset), a.ctypes.data, a.nbytes))
        return self

    def copy_from(self, elem_offset, src, src_offset, count):
        """Device-to-device copy of `count` elements."""
        self.eng.d2d(self.at(elem_offset), src.at(src_offset), int(count) * self.dtype.itemsize)
        return self

    def free(self):
        if self.ptr and getattr(self.eng, "h", None):   # the handle may already be closed
            self.eng.L.pf_dev_free(self.eng.h, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _View(DevBuf):
    """A DevBuf over memory the library owns (never freed here)."""

    def __init__(self, eng, ptr, count, dtype):
        self.eng, self.ptr = eng, ptr
        self.shape, self.dtype = (int(count),), np.dtype(dtype)
        self.nbytes = int(count) * self.dtype.itemsize

    def free(self):
        self.ptr = 0


def score_params(variant=0, restrict_policy=True, w_turn=0.1, w_safe=0.05, min_safe=1.5, diag_pen=1000.0):
    return ScoreParams(int(variant), int(bool(restrict_policy)), float(w_turn), float(w_safe), float(min_safe),
                       float(diag_pen))


class Engine:
    def __init__(self, grid, device=0):
        g = np.ascontiguousarray(np.asarray(grid), dtype=np.int64)
        if g.ndim != 2:
            raise ValueError("grid must be 2-D")
        self.R, self.C = (int(v) for v in g.shape)
        self.grid_u8 = np.ascontiguousarray(np.clip(g, 0, 255).astype(np.uint8))
        self.L = _lib.lib()
        h = C.c_void_p()
        rc = self.L.pf_create(self.grid_u8.ctypes.data, self.R, self.C, int(device), C.byref(h))
        if rc != 0:
            raise PathfitError(self.L.pf_last_error(None).decode())
        self.h = h
        self.device = int(device)
        self._out13 = (C.c_double * 13)()   # pf_maaco_iterate's answer block, reused
        self.klog = None          # a list: every hot-path launch appends (kernel family, its HIP-event ms, its counters)

    def update_grid(self, grid):
        """Dynamic maps: replace the occupancy (same shape); solvers built on the old map must be set up again."""
        g = np.ascontiguousarray(np.asarray(grid), dtype=np.int64)
        if g.shape != (self.R, self.C):
            raise ValueError("update_grid: the new grid must have the handle's shape")
        self.grid_u8 = np.ascontiguousarray(np.clip(g, 0, 255).astype(np.uint8))
        self._ck(self.L.pf_update_grid(self.h, self.grid_u8.ctypes.data))

    def close(self):
        if getattr(self, "h", None):
            self.L.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise PathfitError(self.L.pf_last_error(self.h).decode())

    def _logk(self, name):
        if self.klog is not None:
            self.klog.append((name, self.last_kernel_ms(), self.counters()))

    def buf(self, shape, dtype):
        return DevBuf(self, shape, dtype)

    def put(self, arr, dtype=None):
        a = np.ascontiguousarray(arr, dtype)
        return DevBuf(self, a.shape if a.ndim else (1,), a.dtype).upload(a)

    def d2d(self, dst_ptr, src_ptr, nbytes):
        if nbytes:
            self._ck(self.L.pf_d2d(self.h, dst_ptr, src_ptr, int(nbytes)))

    def set_option(self, name, value):
        self._ck(self.L.pf_set_option(self.h, name.encode(), int(value)))

    def counters(self):
        c = Counters()
        self._ck(self.L.pf_get_counters(self.h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def last_kernel_ms(self):
        return float(self.L.pf_last_kernel_ms(self.h))

    def default_path_cap(self):
        return min(self.R * self.C, 8 * (self.R + self.C) + 64)

    # ------------------------------------------------------------------ K2
    def astar_batch(self, variant, d_start, d_target, n, path_cap, d_cells, d_len, d_status, d_avoid_off=None,
                    d_avoid_cells=None, d_counters=None, allow_diag=True, restrict_corner=True):
        self._ck(self.L.pf_astar_batch(self.h, variant, int(allow_diag), int(restrict_corner), n, d_start.ptr,
                                       d_target.ptr, d_avoid_off.ptr if d_avoid_off else None,
                                       d_avoid_cells.ptr if d_avoid_cells else None, path_cap, d_cells.ptr,
                                       d_len.ptr, d_status.ptr, d_counters.ptr if d_counters else None))
        self._logk("astar")

    def astar_host(self, variant, starts, targets, avoid_lists=None, path_cap=None, allow_diag=True,
                   restrict_corner=True, want_counters=False):
        """-> (list of path arrays, status array[, counters n x 4])."""
        n = len(starts)
        cap = int(path_cap or self.default_path_cap())
        ds, dt = self.put(starts, np.int32), self.put(targets, np.int32)
        dc, dl, dst = self.buf((max(n, 1), cap), np.int32), self.buf(max(n, 1), np.int32), self.buf(max(n, 1), np.int32)
        doff = dav = None
        if avoid_lists is not None:
            off = np.zeros(n + 1, np.int64)
            for i, a in enumerate(avoid_lists):
                off[i + 1] = off[i] + (len(a) if a is not None else 0)
            flat = np.concatenate([np.asarray(a, np.int32) for a in avoid_lists if a is not None and len(a)]) \
                if off[-1] else np.zeros(1, np.int32)
            doff, dav = self.put(off, np.int64), self.put(flat, np.int32)
        dcnt = self.buf((max(n, 1), 4), np.int64) if want_counters else None
        self.astar_batch(variant, ds, dt, n, cap, dc, dl, dst, doff, dav, dcnt, allow_diag, restrict_corner)
        cells, lens, st = dc.download(), dl.download(), dst.download()
        paths = [cells[i, :lens[i]].copy() for i in range(n)]
        if want_counters:
            return paths, st[:n], dcnt.download()[:n]
        return paths, st[:n]

    # ------------------------------------------------------------------ K1
    def score_host(self, paths, sp):
        n = len(paths)
        cap = max([len(p) for p in paths] + [1])
        cells = np.zeros((n, cap), np.int32)
        lens = np.zeros(n, np.int32)
        for i, p in enumerate(paths):
            cells[i, :len(p)] = p
            lens[i] = len(p)
        dc, dl, ds = self.put(cells), self.put(lens), self.buf((n, 5), np.float64)
        self._ck(self.L.pf_score_batch(self.h, C.byref(sp), n, cap, dc.ptr, dl.ptr, ds.ptr))
        return ds.download()

    # ------------------------------------------------------------------ K3
    def decode_batch(self, n, W, start, target, path_cap, d_cells, d_len, d_status, d_wp_cells=None, d_wp_pos=None,
                     sp=None, d_stats=None, allow_diag=True, restrict_corner=True):
        self._ck(self.L.pf_decode_batch(self.h, int(allow_diag), int(restrict_corner), n, W,
                                        d_wp_cells.ptr if d_wp_cells else None, d_wp_pos.ptr if d_wp_pos else None,
                                        int(start), int(target), path_cap, d_cells.ptr, d_len.ptr, d_status.ptr,
                                        C.byref(sp) if sp is not None else None, d_stats.ptr if d_stats else None))
        self._logk("decode")

    def decode_host(self, start, target, wp_cells=None, wp_pos=None, sp=None, path_cap=None, allow_diag=True,
                    restrict_corner=True):
        """wp_cells int[n, W] or wp_pos float[n, W, 2] -> (paths, status, stats or None)."""
        if wp_cells is not None:
            wp = np.ascontiguousarray(wp_cells, np.int32)
            n, W = wp.shape
            dwc, dwp = self.put(wp.reshape(-1) if wp.size else np.zeros(1, np.int32)), None
        else:
            wp = np.ascontiguousarray(wp_pos, np.float64)
            n, W = wp.shape[0], wp.shape[1]
            dwc, dwp = None, self.put(wp.reshape(-1) if wp.size else np.zeros(1))
        cap = int(path_cap or self.default_path_cap())
        dc, dl, dst = self.buf((n, cap), np.int32), self.buf(n, np.int32), self.buf(n, np.int32)
        dstat = self.buf((n, 5), np.float64) if sp is not None else None
        self.decode_batch(n, W, start, target, cap, dc, dl, dst, dwc, dwp, sp, dstat, allow_diag, restrict_corner)
        cells, lens, st = dc.download(), dl.download(), dst.download()
        paths = [cells[i, :lens[i]].copy() for i in range(n)]
        return paths, st, (dstat.download() if dstat is not None else None)

    # ------------------------------------------------------------------ K6
    def pso_update(self, n, W, w, c1, c2, max_vel, d_pos, d_vel, d_pbest, d_gbest, seed, it, agent0=0):
        self._ck(self.L.pf_pso_update(self.h, n, W, w, c1, c2, max_vel, d_pos.ptr, d_vel.ptr, d_pbest.ptr,
                                      d_gbest.ptr, int(seed), int(it), int(agent0)))

    def pso_update_raw(self, n, W, w, c1, c2, max_vel, pos_ptr, vel_ptr, pbest_ptr, gbest_ptr, seed, it, agent0):
        self._ck(self.L.pf_pso_update(self.h, n, W, w, c1, c2, max_vel, pos_ptr, vel_ptr, pbest_ptr, gbest_ptr,
                                      int(seed), int(it), int(agent0)))

    def pso_update_keep_raw(self, n, W, w, c1, c2, max_vel, pos_ptr, vel_ptr, pbest_ptr, gbest_ptr, seed, it, agent0, pos_keep_ptr, vel_keep_ptr):
        """pso_update_raw + the pre-update position / velocity kept for the roll-back; nothing synchronises."""
        self._ck(self.L.pf_pso_update_keep(self.h, n, W, w, c1, c2, max_vel, pos_ptr, vel_ptr, pbest_ptr, gbest_ptr,
                                           int(seed), int(it), int(agent0), pos_keep_ptr, vel_keep_ptr))

    def pso_commit_raw(self, m, W, path_cap, n_final, improver, pos_ptr, vel_ptr, pos_keep_ptr, vel_keep_ptr, stats_ptr, len_ptr, cells_ptr,
                       pbest_ptr, pbf_ptr, pb_cells_ptr, pb_len_ptr, gb_ptr, gstats_ptr, gpath_ptr):
        """One round of the asynchronous sweep committed in one launch (pbest + paths, gbest move, roll-back); nothing synchronises."""
        self._ck(self.L.pf_pso_commit(self.h, int(m), int(W), int(path_cap), int(n_final), int(improver), pos_ptr, vel_ptr, pos_keep_ptr,
                                      vel_keep_ptr, stats_ptr, len_ptr, cells_ptr, pbest_ptr, pbf_ptr, pb_cells_ptr, pb_len_ptr, gb_ptr,
                                      gstats_ptr, gpath_ptr))

    def decode_raw(self, n, W, start, target, path_cap, cells_ptr, len_ptr, status_ptr, wp_pos_ptr, sp, stats_ptr,
                   allow_diag=True, restrict_corner=True):
        self._ck(self.L.pf_decode_batch(self.h, int(allow_diag), int(restrict_corner), n, W, None, wp_pos_ptr,
                                        int(start), int(target), path_cap, cells_ptr, len_ptr, status_ptr,
                                        C.byref(sp), stats_ptr))
        self._logk("decode")

    def pso_pbest_raw(self, n, W, pos_ptr, stats_ptr, len_ptr, pbest_ptr, pbf_ptr, imp_ptr):
        self._ck(self.L.pf_pso_pbest(self.h, n, W, pos_ptr, stats_ptr, len_ptr, pbest_ptr, pbf_ptr, imp_ptr))

    def pso_scan(self, n, stats_ptr, len_ptr, status_ptr, pbf_ptr, gbest_fit, sync_mode):
        """-> (index of the gbest improver in the batch or -1, its fitness, overflowed particles); one 16-byte D2H."""
        i, f, o = C.c_int32(-1), C.c_double(INF), C.c_int32(0)
        self._ck(self.L.pf_pso_scan(self.h, n, stats_ptr, len_ptr, status_ptr, pbf_ptr, float(gbest_fit), int(sync_mode),
                                    C.byref(i), C.byref(f), C.byref(o)))
        return i.value, f.value, o.value

    def pso_pbest_paths_raw(self, n, path_cap, cells_ptr, len_ptr, imp_ptr, pb_cells_ptr, pb_len_ptr):
        self._ck(self.L.pf_pso_pbest_paths(self.h, n, path_cap, cells_ptr, len_ptr, imp_ptr, pb_cells_ptr, pb_len_ptr))

    def d2h_counts(self):
        """(small copies <= 64 B, bulk copies, bulk bytes) this handle has made device -> host."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._ck(self.L.pf_d2h_counts(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def span_begin(self):
        """Open a HIP-event timed span on the engine's stream (closed by span_end; not nested, nothing synchronises)."""
        self._ck(self.L.pf_span_begin(self.h))

    def span_end(self):
        self._ck(self.L.pf_span_end(self.h))

    def span_total(self, reset=True):
        """(milliseconds, spans) summed since the last reset; synchronises the stream."""
        ms, cnt = C.c_double(0.0), C.c_int64(0)
        self._ck(self.L.pf_span_total(self.h, C.byref(ms), C.byref(cnt), 1 if reset else 0))
        return ms.value, cnt.value

    def read(self, ptr, count, dtype):
        """A small typed read from device memory (row reads on demand)."""
        out = np.empty(int(count), dtype)
        if out.nbytes:
            self._ck(self.L.pf_d2h(self.h, out.ctypes.data, ptr, out.nbytes))
        return out

    def pso_pbest(self, n, W, d_pos, d_stats, d_len, d_pbest, d_pbest_fit, d_improved):
        self._ck(self.L.pf_pso_pbest(self.h, n, W, d_pos.ptr, d_stats.ptr, d_len.ptr, d_pbest.ptr, d_pbest_fit.ptr,
                                     d_improved.ptr))

    # ------------------------------------------------------------------ K4/K5
    def maaco_setup(self, params):
        self._mp = params
        self._ck(self.L.pf_maaco_setup(self.h, C.byref(params)))

    def maaco_walk(self, it, seed, ant0, n, path_cap, d_cells, d_len, d_plen, d_turns, d_status):
        self._ck(self.L.pf_maaco_walk_batch(self.h, int(it), int(seed), int(ant0), n, path_cap, d_cells.ptr,
                                            d_len.ptr, d_plen.ptr, d_turns.ptr, d_status.ptr))
        self._logk("maaco_walk")

    def maaco_update(self, n, path_cap, d_cells, d_len, d_plen, best_len_overall):
        """MAACO.py:304-332 in one pass over tau (evaporate, ordered deposits, clip): the single-GPU form of the three calls below."""
        self._ck(self.L.pf_maaco_update(self.h, n, path_cap, d_cells.ptr, d_len.ptr, d_plen.ptr, float(best_len_overall)))

    def maaco_iterate(self, it, seed, ant0, n, path_cap, d_cells, d_len, d_plen, d_turns, d_status, best_len, best_turns):
        """One whole iteration (walks, best scan, take-over test, pheromone update) enqueued back to back; one 104-byte block back
        (mirrored by the device into pinned host memory: the call returns once the take-over test is known).
        -> dict(ib_len, ib_turns, ib_idx, took, best_len, best_turns, skipped, overflow_agents)."""
        out = self._out13
        self._ck(self.L.pf_maaco_iterate(self.h, int(it), int(seed), int(ant0), int(n), int(path_cap), d_cells.ptr, d_len.ptr, d_plen.ptr,
                                         d_turns.ptr, d_status.ptr, float(best_len), float(best_turns), C.addressof(out)))
        self._logk("maaco_walk")
        return {"ib_len": float(out[0]), "ib_turns": float(out[1]), "ib_idx": int(out[2]), "took": out[3] != 0.0, "best_len": float(out[4]),
                "best_turns": float(out[5]), "skipped": out[8] != 0.0, "overflow_agents": int(out[12])}

    def maaco_best_path(self, cap):
        """The overall best ant's cells as pf_maaco_iterate keeps them in HBM (empty: none yet)."""
        out = np.empty(int(cap), np.int32)
        L = C.c_int32(0)
        self._ck(self.L.pf_maaco_best_path(self.h, out.ctypes.data, int(cap), C.byref(L)))
        return out[:L.value].copy()

    def maaco_evaporate(self):
        self._ck(self.L.pf_maaco_evaporate(self.h))

    def maaco_deposit(self, n, path_cap, d_cells, d_len, d_plen):
        self._ck(self.L.pf_maaco_deposit(self.h, n, path_cap, d_cells.ptr, d_len.ptr, d_plen.ptr))

    def maaco_deposit_begin(self, n, path_cap, d_cells, d_len, d_plen):
        self._ck(self.L.pf_maaco_deposit_begin(self.h, n, path_cap, d_cells.ptr, d_len.ptr, d_plen.ptr))

    def maaco_deposit_cells(self, cell0, cell1):
        self._ck(self.L.pf_maaco_deposit_cells(self.h, int(cell0), int(cell1)))

    def maaco_best_dev(self, n, d_plen, d_turns):
        """MAACO.py:343-349 over device columns -> (best_len, best_turns, best_idx); one 24-byte D2H."""
        out = np.zeros(3)
        self._ck(self.L.pf_maaco_best_dev(self.h, int(n), d_plen.ptr, d_turns.ptr, out.ctypes.data))
        return float(out[0]), float(out[1]), int(out[2])

    @property
    def tau_buf(self):
        """The pheromone matrix in HBM as a buffer object (R*C doubles, owned by the library)."""
        return _View(self, self.L.pf_maaco_tau_dev(self.h), self.R * self.C, np.float64)

    def maaco_clip(self, best_len_overall):
        self._ck(self.L.pf_maaco_clip(self.h, float(best_len_overall)))

    def maaco_get_pheromone(self):
        t = np.empty((self.R, self.C), np.float64)
        self._ck(self.L.pf_maaco_get_pheromone(self.h, t.ctypes.data))
        return t

    def maaco_set_pheromone(self, tau):
        t = np.ascontiguousarray(tau, np.float64)
        assert t.size == self.R * self.C
        self._ck(self.L.pf_maaco_set_pheromone(self.h, t.ctypes.data))

    def maaco_tau_ptr(self):
        return self.L.pf_maaco_tau_dev(self.h)

    def maaco_best_scan(self, plen, turns, idx0, best_len, best_turns, best_idx):
        plen = np.ascontiguousarray(plen, np.float64)
        turns = np.ascontiguousarray(turns, np.int32)
        bl, bt, bi = C.c_double(best_len), C.c_double(best_turns), C.c_int32(best_idx)
        self._ck(self.L.pf_maaco_best_scan(len(plen), plen.ctypes.data, turns.ctypes.data, int(idx0), C.byref(bl),
                                           C.byref(bt), C.byref(bi)))
        return bl.value, bt.value, bi.value

    # ------------------------------------------------------------------ K7
    def mpa_setup(self, mp, sp):
        self._ck(self.L.pf_mpa_setup(self.h, C.byref(mp), C.byref(sp)))

    def mpa_phase(self, phase, CF, it, seed, n, path_cap, d_pop_cells, d_pop_len, d_pop_stats, d_gidx, d_slot,
                  elite_cells_ptr, elite_len, elite_stats_ptr, d_out_cells, d_out_len, d_out_stats, d_status):
        self._ck(self.L.pf_mpa_phase_batch(self.h, int(phase), float(CF), int(it), int(seed), n, path_cap,
                                           d_pop_cells.ptr, d_pop_len.ptr, d_pop_stats.ptr, d_gidx.ptr, d_slot.ptr,
                                           elite_cells_ptr, int(elite_len), elite_stats_ptr, d_out_cells.ptr,
                                           d_out_len.ptr, d_out_stats.ptr, d_status.ptr))

    def mpa_iter(self, phase, CF, it, seed, n, path_cap, d_pop_cells, d_pop_len, d_pop_stats, d_gidx, d_slot,
                 elite_cells_ptr, elite_len, elite_stats_ptr, d_c1_cells, d_c1_len, d_c1_stats, d_c2_cells, d_c2_len,
                 d_c2_stats, d_status):
        self._ck(self.L.pf_mpa_iter_batch(self.h, int(phase), float(CF), int(it), int(seed), n, path_cap,
                                          d_pop_cells.ptr, d_pop_len.ptr, d_pop_stats.ptr, d_gidx.ptr, d_slot.ptr,
                                          elite_cells_ptr, int(elite_len), elite_stats_ptr, d_c1_cells.ptr,
                                          d_c1_len.ptr, d_c1_stats.ptr, d_c2_cells.ptr, d_c2_len.ptr, d_c2_stats.ptr,
                                          d_status.ptr))
        self._logk("mpa_sweep")

    # ------------------------------------------------------------------ GA generation in HBM
    def ga_select(self, seed, gen, n, k, d_fit_all, d_gorder, d_psid):
        self._ck(self.L.pf_ga_select_dev(self.h, int(seed), int(gen), int(n), int(k), d_fit_all.ptr, d_gorder.ptr, d_psid.ptr))

    def ga_breed(self, seed, gen, N, W, cx, mut, d_chrom_all, d_psid, child0, nchild, d_out):
        self._ck(self.L.pf_ga_breed_dev(self.h, int(seed), int(gen), int(N), int(W), float(cx), float(mut), d_chrom_all.ptr, d_psid.ptr,
                                        int(child0), int(nchild), d_out.ptr))

    def ga_assemble(self, n_loc, W, cap, lo, kid_len, kid_chrom, kid_stats, kid_cells, psid, chrom_old, stats_old, cells_old, len_old,
                    old_lo, old_hi, chrom_new, stats_new, cells_new, len_new):
        self._ck(self.L.pf_ga_assemble_dev(self.h, int(n_loc), int(W), int(cap), int(lo), kid_len.ptr, kid_chrom.ptr, kid_stats.ptr,
                                           kid_cells.ptr, psid.ptr, chrom_old.ptr, stats_old.ptr, cells_old.ptr, len_old.ptr,
                                           int(old_lo), int(old_hi), chrom_new.ptr, stats_new.ptr, cells_new.ptr, len_new.ptr))

    def sort_order_by_key(self, n, d_vals, stride, offset, d_order):
        self._ck(self.L.pf_sort_order_by_key(self.h, int(n), d_vals.ptr, int(stride), int(offset), d_order.ptr))

    def sorted_head(self, d_order, d_vals, stride=1, offset=0):
        """(id at the head of the sorted list, its key): one 16-byte copy."""
        out = np.zeros(2)
        self._ck(self.L.pf_sorted_head(self.h, d_order.ptr, d_vals.ptr, int(stride), int(offset), out.ctypes.data))
        return int(out[0]), float(out[1])

    def vec_add(self, n, d_a, d_b, sign, d_out):
        """d_out = d_a + sign * d_b (sign +1 / -1) on device columns."""
        self._ck(self.L.pf_vec_add_f64(self.h, int(n), d_a.ptr, d_b.ptr, float(sign), d_out.ptr))

    def gather_col(self, n, d_src, stride, offset, d_dst):
        self._ck(self.L.pf_gather_col(self.h, int(n), d_src.ptr, int(stride), int(offset), d_dst.ptr))

    def mpa_elite_bufs(self):
        """(cells, len, stats) views of the library's elite buffer."""
        c, l, s_ = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._ck(self.L.pf_mpa_elite_buf(self.h, C.byref(c), C.byref(l), C.byref(s_)))
        return (_View(self, c.value, self.R * self.C, np.int32), _View(self, l.value, 1, np.int32), _View(self, s_.value, 5, np.float64))

    def mpa_pick_elite(self, path_cap, d_pop_cells, d_pop_len, d_pop_stats, d_order, first_id=0):
        self._ck(self.L.pf_mpa_pick_elite(self.h, int(path_cap), d_pop_cells.ptr, d_pop_len.ptr, d_pop_stats.ptr, d_order.ptr, int(first_id)))

    def mpa_local_view(self, N, d_gorder, lo, hi, d_gidx, d_slot):
        self._ck(self.L.pf_mpa_local_view(self.h, int(N), d_gorder.ptr, int(lo), int(hi), d_gidx.ptr, d_slot.ptr))

    def mpa_fads(self, CF, it, seed, n, path_cap, d_gidx, d_slot, d_pop_cells, d_pop_len, d_pop_stats, d_status):
        self._ck(self.L.pf_mpa_fads_batch(self.h, float(CF), int(it), int(seed), n, path_cap, d_gidx.ptr, d_slot.ptr,
                                          d_pop_cells.ptr, d_pop_len.ptr, d_pop_stats.ptr, d_status.ptr))

    def mpa_memory(self, n, path_cap, d_slot, d_cand_cells, d_cand_len, d_cand_stats, d_pop_cells, d_pop_len,
                   d_pop_stats):
        self._ck(self.L.pf_mpa_memory(self.h, n, path_cap, d_slot.ptr, d_cand_cells.ptr, d_cand_len.ptr,
                                      d_cand_stats.ptr, d_pop_cells.ptr, d_pop_len.ptr, d_pop_stats.ptr))

"""ctypes binding of the CPU oracle (oracle/pf_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INF = float("inf")


def build(force=False):
    so = os.path.join(_HERE, "libpf_oracle.so")
    src = os.path.join(_HERE, "pf_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpf_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class MaacoParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("alpha", "beta", "rho", "Q", "a_turn", "wh_max", "wh_min", "k_h", "q0_initial", "C0")] + \
               [("num_iterations", C.c_int)]


class Rng(C.Structure):
    _fields_ = [("key", C.c_uint64), ("ctr", C.c_uint64)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
        L.orc_ws_create.restype = vp; L.orc_ws_create.argtypes = [i32, i32]
        L.orc_ws_destroy.argtypes = [vp]
        L.orc_astar.restype = i64
        L.orc_astar.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, i64, vp]
        L.orc_score.argtypes = [vp, i32, i32, vp, i64, i32, dbl, dbl, dbl, i32, dbl, i32, vp]
        L.orc_pso_round.argtypes = [vp, i32, i32, i32, vp]
        L.orc_decode.restype = i64
        L.orc_decode.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, vp, i64, vp]
        L.orc_pso_update.argtypes = [i32, i32, dbl, dbl, dbl, dbl, i32, i32, vp, vp, vp, vp, u64, u64, u64]
        L.orc_maaco_init_pheromone.argtypes = [vp, i32, i32, i32, i32, dbl, vp]
        L.orc_maaco_dist_to_target.argtypes = [i32, i32, i32, vp]
        L.orc_maaco_q0.restype = dbl; L.orc_maaco_q0.argtypes = [i32, i32, dbl]
        L.orc_maaco_walk.restype = i64
        L.orc_maaco_walk.argtypes = [vp, i32, i32, i32, i32, C.POINTER(MaacoParams), vp, vp, i32, u64, u64, vp,
                                     vp, i64, vp, vp, vp]
        L.orc_maaco_update.argtypes = [vp, i32, i32, dbl, dbl, vp, i32, vp, vp, vp, dbl]
        L.orc_rng_init.argtypes = [C.POINTER(Rng), u64, u64, u64, u64]
        L.orc_rng_next64.restype = u64; L.orc_rng_next64.argtypes = [C.POINTER(Rng)]
        L.orc_rng_random.restype = dbl; L.orc_rng_random.argtypes = [C.POINTER(Rng)]
        L.orc_rng_randbelow.restype = u64; L.orc_rng_randbelow.argtypes = [C.POINTER(Rng), u64]
        L.orc_rng_randint.restype = i64; L.orc_rng_randint.argtypes = [C.POINTER(Rng), i64, i64]
        L.orc_rng_uniform.restype = dbl; L.orc_rng_uniform.argtypes = [C.POINTER(Rng), dbl, dbl]
        L.orc_rng_normalvariate.restype = dbl; L.orc_rng_normalvariate.argtypes = [C.POINTER(Rng), dbl, dbl]
        L.orc_mpa_levy_target.restype = i32
        L.orc_mpa_levy_target.argtypes = [C.POINTER(Rng), i32, i32, i32, dbl, dbl, dbl]
        L.orc_mpa_brownian_target.restype = i32
        L.orc_mpa_brownian_target.argtypes = [C.POINTER(Rng), i32, i32, i32, i32, dbl]
        L.orc_mpa_rebuild.restype = i64
        L.orc_mpa_rebuild.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i64, vp, i64, i64, i32, dbl, dbl, dbl,
                                      C.POINTER(Rng), vp, vp, i64, vp, vp]
        L.orc_mpa_targets_batch.argtypes = [u64, i64, i32, i32, i32, vp, vp, dbl, dbl, dbl, vp]
        L.orc_set_step_cap.argtypes = [i64]
        _LIB = L
    return _LIB


def set_step_cap(cap):
    """Test hook: lower both connectors' step cap (0 = the reference's 3RC / 2RC)."""
    lib().orc_set_step_cap(int(cap))


def mpa_targets_batch(seed, is_levy, R, Cc, cur, elite, scale, levy_beta, sigma):
    cur = np.ascontiguousarray(cur, np.int32); elite = np.ascontiguousarray(elite, np.int32)
    out = np.zeros(cur.size, np.int32)
    lib().orc_mpa_targets_batch(int(seed), cur.size, int(is_levy), int(R), int(Cc), _p(cur), _p(elite), float(scale),
                                float(levy_beta), float(sigma), _p(out))
    return out


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def occ_of(grid):
    """uint8 occupancy (1 = obstacle) from any reference-style grid (env.py:4-7)."""
    g = np.asarray(grid)
    return np.ascontiguousarray((g == 1).astype(np.uint8))


class Oracle:
    """One grid + reusable workspace."""

    def __init__(self, grid, allow_diag=True, restrict_corner=True):
        self.occ = occ_of(grid)
        self.R, self.C = self.occ.shape
        self.allow_diag, self.restrict = int(allow_diag), int(restrict_corner)
        self.L = lib()
        self.ws = self.L.orc_ws_create(self.R, self.C)
        self._buf = np.empty(self.R * self.C, np.int32)
        self._scratch = np.zeros(self.R * self.C, np.uint8)

    def __del__(self):
        try:
            self.L.orc_ws_destroy(self.ws)
        except Exception:
            pass

    def cell(self, rc):
        return int(rc[0]) * self.C + int(rc[1])

    def rc(self, cell):
        return (int(cell) // self.C, int(cell) % self.C)

    def avoid_mask(self, cells):
        m = np.zeros(self.R * self.C, np.uint8)
        if cells is not None and len(cells):
            m[np.asarray(list(cells), np.int64)] = 1
        return m

    def astar(self, start, target, avoid=None, variant=0):
        """-> (path cells ndarray, stats[6])."""
        st = np.zeros(6, np.int64)
        am = avoid if isinstance(avoid, np.ndarray) and avoid.dtype == np.uint8 else \
            (self.avoid_mask(avoid) if avoid is not None else None)
        n = self.L.orc_astar(self.ws, _p(self.occ), self.R, self.C, variant, self.allow_diag, self.restrict,
                             int(start), int(target), _p(am), _p(self._buf), self._buf.size, _p(st))
        assert n >= 0
        return self._buf[:n].copy(), st

    def score(self, path, variant=0, w_turn=0.1, w_safe=0.05, min_safe=1.5, restrict_policy=True, diag_pen=1000.0,
              literal_safety=False):
        path = np.ascontiguousarray(path, np.int32)
        out = np.zeros(5)
        self.L.orc_score(_p(self.occ), self.R, self.C, _p(path), path.size, variant, w_turn, w_safe, min_safe,
                         int(restrict_policy), diag_pen, int(literal_safety), _p(out))
        return out

    def pso_round(self, pos):
        pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 2)
        cells = np.zeros(pos.shape[0], np.int32)
        self.L.orc_pso_round(_p(pos), pos.shape[0], self.R, self.C, _p(cells))
        return cells

    def decode(self, start, target, wps):
        wps = np.ascontiguousarray(wps, np.int32)
        st = np.zeros(6, np.int64)
        n = self.L.orc_decode(self.ws, _p(self.occ), self.R, self.C, self.allow_diag, self.restrict, int(start),
                              int(target), _p(wps), wps.size, _p(self._scratch), _p(self._buf), self._buf.size, _p(st))
        assert n >= 0
        return self._buf[:n].copy(), st

    def pso_update(self, pos, vel, pbest, gbest, w, c1, c2, max_vel, seed, it, agent0=0):
        pos = np.array(pos, np.float64, order="C"); vel = np.array(vel, np.float64, order="C")
        pbest = np.ascontiguousarray(pbest, np.float64); gbest = np.ascontiguousarray(gbest, np.float64)
        n, W = pos.shape[0], pos.shape[1]
        self.L.orc_pso_update(n, W, w, c1, c2, max_vel, self.R, self.C, _p(pos), _p(vel), _p(pbest), _p(gbest),
                              seed, it, agent0)
        return pos, vel

    # ---- MAACO ----
    def maaco_init(self, start, target, C0):
        tau = np.zeros(self.R * self.C)
        dist = np.zeros(self.R * self.C)
        self.L.orc_maaco_init_pheromone(_p(self.occ), self.R, self.C, int(start), int(target), C0, _p(tau))
        self.L.orc_maaco_dist_to_target(self.R, self.C, int(target), _p(dist))
        return tau, dist

    def maaco_q0(self, it, K, q0_initial):
        return self.L.orc_maaco_q0(it, K, q0_initial)

    def maaco_walk(self, start, target, params, tau, dist, it, seed, ant):
        out_len = C.c_double(INF); out_turns = C.c_int64(0)
        cnt = np.zeros(5, np.int64)
        n = self.L.orc_maaco_walk(_p(self.occ), self.R, self.C, int(start), int(target), C.byref(params), _p(tau),
                                  _p(dist), it, seed, ant, _p(self._scratch), _p(self._buf), self._buf.size,
                                  C.byref(out_len), C.byref(out_turns), _p(cnt))
        assert n >= 0
        if n == 0:
            return np.zeros(0, np.int32), INF, INF, cnt
        return self._buf[:n].copy(), out_len.value, out_turns.value, cnt

    def maaco_update(self, tau, rho, Q, paths, lens, best_len_overall):
        offs = np.zeros(len(paths) + 1, np.int64)
        for i, p in enumerate(paths):
            offs[i + 1] = offs[i] + len(p)
        cells = np.concatenate([np.asarray(p, np.int32) for p in paths]) if offs[-1] else np.zeros(0, np.int32)
        cells = np.ascontiguousarray(cells, np.int32)
        lens = np.ascontiguousarray(lens, np.float64)
        self.L.orc_maaco_update(_p(self.occ), self.R, self.C, rho, Q, _p(tau), len(paths), _p(offs), _p(cells),
                                _p(lens), best_len_overall)
        return tau

    # ---- MPA ----
    def rng(self, seed, dom, it, agent):
        g = Rng()
        self.L.orc_rng_init(C.byref(g), seed, dom, it, agent)
        return g

    def mpa_rebuild(self, start, target, path, elite, idx, is_levy, scale, levy_beta, sigma, g):
        path = np.ascontiguousarray(path, np.int32); elite = np.ascontiguousarray(elite, np.int32)
        st = np.zeros(6, np.int64); tc = C.c_int(-1)
        n = self.L.orc_mpa_rebuild(self.ws, _p(self.occ), self.R, self.C, self.allow_diag, self.restrict, int(start),
                                   int(target), _p(path), path.size, _p(elite), elite.size, int(idx), int(is_levy),
                                   scale, levy_beta, sigma, C.byref(g), _p(self._scratch), _p(self._buf),
                                   self._buf.size, C.byref(tc), _p(st))
        assert n != -1
        if n == -2:
            return path.copy(), False, tc.value, st
        return self._buf[:n].copy(), True, tc.value, st

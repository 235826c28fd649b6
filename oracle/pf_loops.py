"""Solver outer loops driven by the CPU oracle (TEST INFRASTRUCTURE).

These restate the *host* control flow of the reference's solve loops
(MAACO.py:334-371, MPA.py:320-448, pso.py:163-240) around the oracle's
per-agent functions, with the engine's per-agent stream keys, so that
  * tests can check the GPU facades end to end on the GPU box, and
  * bench.py can time a bounded CPU sample (cpu_baseline, kind "port").
"""
import ctypes as C
import math
import time

import numpy as np

import pf_oracle as po

INF = float("inf")
DOM_MAACO, DOM_MPA, DOM_PSO, DOM_MPA_FADS = 1, 2, 3, 5


def levy_sigma(beta):
    num = math.gamma(1 + beta) * math.sin(math.pi * beta / 2)
    den = math.gamma((1 + beta) / 2) * beta * (2 ** ((beta - 1) / 2))
    return (num / den) ** (1 / beta) if den > 1e-9 else 1.0


# --------------------------------------------------------------------------- MAACO
def maaco_solve(orc, start, target, num_ants, num_iterations, alpha, beta, rho, Q, a_turn_coef, wh_max, wh_min,
                k_h_adaptive, q0_initial, C0=0.1, seed=0, iters=None):
    P = po.MaacoParams(alpha=alpha, beta=beta, rho=rho, Q=Q, a_turn=a_turn_coef, wh_max=wh_max, wh_min=wh_min,
                       k_h=k_h_adaptive, q0_initial=q0_initial, C0=C0, num_iterations=num_iterations)
    tau, dist = orc.maaco_init(start, target, C0)
    best_path, best_len, best_turns, curve = np.zeros(0, np.int32), INF, INF, []
    for it in range(1, (iters or num_iterations) + 1):
        paths, lens = [], []
        ib_len, ib_turns, ib_path = INF, INF, np.zeros(0, np.int32)
        for ant in range(num_ants):
            p, L, T, _ = orc.maaco_walk(start, target, P, tau, dist, it, seed, ant)
            paths.append(p); lens.append(L)
            if L < ib_len:                                          # MAACO.py:343-349
                ib_len, ib_path, ib_turns = L, p, T
            elif abs(L - ib_len) < 1e-9 and T < ib_turns:
                ib_path, ib_turns = p, T
        if ib_len < best_len:                                       # :351-358
            best_len, best_path, best_turns = ib_len, ib_path, ib_turns
        elif abs(ib_len - best_len) < 1e-9 and ib_turns < best_turns:
            best_path, best_turns = ib_path, ib_turns
        orc.maaco_update(tau, rho, Q, paths, lens, best_len)       # :359
        curve.append(best_len if best_len != INF else None)
    return dict(path=best_path, length=best_len, turns=best_turns, curve=curve, tau=tau.reshape(orc.R, orc.C))


# --------------------------------------------------------------------------- MPA
class MpaOracle:
    """MPA.solve_path_planning (MPA.py:320-448) over the oracle."""

    def __init__(self, orc, start, target, num_predators, num_iterations, FADs_rate=0.2, P_const=0.5, levy_beta=1.5,
                 w_turn=0.1, w_safe=0.05, min_safe=1.5, diag_pen=1000.0, restrict=True, seed=0):
        self.o, self.s, self.t = orc, start, target
        self.N, self.K = num_predators, num_iterations
        self.fads_rate, self.P, self.beta, self.sigma = FADs_rate, P_const, levy_beta, levy_sigma(levy_beta)
        self.sw = (w_turn, w_safe, min_safe, restrict, diag_pen)
        self.seed = seed
        p, _ = orc.astar(start, target, None, 1)                    # MPA.py:231-245
        if len(p) == 0:
            p = np.array([start, target] if orc.occ.reshape(-1)[target] != 1 else [start], np.int32)
        st = self.score(p)
        self.pop = [(p.copy(), st.copy()) for _ in range(self.N)]
        self.best = None
        self.curve = []
        self.L = po.lib()

    def score(self, p):
        w = self.sw
        return self.o.score(p, 1, w[0], w[1], w[2], w[3], w[4])

    def _sort(self):
        self.pop.sort(key=lambda x: x[1][4])

    def phase_candidate(self, it, i, elite, CF):
        """One predator of the phase sweep MPA.py:339-377 -> (path, stats)."""
        L, o = self.L, self.o
        prey_p, prey_s = self.pop[i]
        phase = 1 if it <= self.K / 3 else (2 if it <= 2 * self.K / 3 else 3)
        if phase == 1:
            is_levy, scale, mod, mod_s, ref = False, self.P, prey_p, prey_s, elite[0]
        elif phase == 2:
            is_levy = i < self.N // 2
            scale = self.P if is_levy else self.P * CF
            mod, mod_s = (prey_p, prey_s) if is_levy else elite
            ref = elite[0] if is_levy else prey_p
        else:
            is_levy, scale, mod, mod_s, ref = True, self.P * CF, elite[0], elite[1], prey_p
        gate_p = self.P if phase == 1 else scale
        if len(mod) <= 1:
            return mod, mod_s
        g = o.rng(self.seed, DOM_MPA, it, i)
        idx = L.orc_rng_randint(C.byref(g), 0, len(mod) - 2)
        if L.orc_rng_random(C.byref(g)) < gate_p:
            out, isnew, _, _ = o.mpa_rebuild(self.s, self.t, mod, ref, idx, is_levy, scale, self.beta, self.sigma, g)
            return (out, self.score(out)) if isnew else (mod, mod_s)
        return mod, mod_s

    def fads(self, it, i, ind, CF):
        """MPA.py:387-410 for predator i."""
        L, o = self.L, self.o
        g = o.rng(self.seed, DOM_MPA_FADS, it, i)
        if L.orc_rng_random(C.byref(g)) < self.fads_rate:
            if L.orc_rng_random(C.byref(g)) < CF:
                r = L.orc_rng_randint(C.byref(g), 0, o.R - 1)
                c = L.orc_rng_randint(C.byref(g), 0, o.C - 1)
                node = r * o.C + c
                if o.occ[r, c] != 1:
                    p1, _ = o.astar(self.s, node, None, 1)
                    if len(p1):
                        p2, _ = o.astar(node, self.t, p1[:-1], 1)
                        if len(p2):
                            raw = np.concatenate([p1, p2[1:]])
                            keep = np.ones(len(raw), bool); keep[1:] = raw[1:] != raw[:-1]
                            uq = raw[keep]
                            if len(uq) and uq[-1] == self.t:
                                st = self.score(uq)
                                if st[4] < ind[1][4]:
                                    return (uq, st)
            else:
                p, _ = o.astar(self.s, self.t, None, 1)
                if len(p):
                    st = self.score(p)
                    if st[4] < ind[1][4]:
                        return (p, st)
        return ind

    def step(self, it, predators=None):
        self._sort()                                                 # :333
        elite = self.pop[0]
        ratio = it / self.K
        CF = 0.0 if ratio >= 1.0 else ((1.0 - ratio) ** (2.0 * ratio) if ratio > 0 else 1.0)
        ids = range(self.N) if predators is None else predators
        cand = {i: self.phase_candidate(it, i, elite, CF) for i in ids}
        newpop = list(self.pop)
        for i in ids:                                                # memory :381-384
            if cand[i][1][4] < self.pop[i][1][4]:
                newpop[i] = cand[i]
        for i in ids:                                                # FADs :387-410
            newpop[i] = self.fads(it, i, newpop[i], CF)
        self.pop = newpop
        self._sort()                                                 # :412
        cur = self.pop[0]
        b = self.best
        if b is None or cur[1][4] < b[1][4]:
            self.best = cur
        elif abs(cur[1][4] - b[1][4]) < 1e-9:                        # :422-437
            c, s = cur[1], b[1]
            if c[0] < s[0] or (abs(c[0] - s[0]) < 1e-9 and c[1] < s[1]) or \
               (abs(c[0] - s[0]) < 1e-9 and abs(c[1] - s[1]) < 1e-9 and c[2] < s[2]) or \
               (abs(c[0] - s[0]) < 1e-9 and abs(c[1] - s[1]) < 1e-9 and abs(c[2] - s[2]) < 1e-9 and c[3] < s[3]):
                self.best = cur
        self.curve.append(self.best[1][4])

    def solve(self):
        self._sort()
        self.best = self.pop[0]
        self.curve.append(self.best[1][4])
        for it in range(1, self.K + 1):
            self.step(it)
        return self.best


# --------------------------------------------------------------------------- timing helper
def timed(fn, *a, **k):
    t0 = time.perf_counter()
    r = fn(*a, **k)
    return r, time.perf_counter() - t0

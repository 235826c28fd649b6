#!/usr/bin/env python3
"""CPU study for DESIGN.md 4.3: the closed-set connector's path as a function of a settled g-field.

    python scripts/settle/run_study.py [n_cases]

Builds scripts/settle/settle_study.c into /tmp, runs the model next to the oracle's sequential restatement on random
pairs and on chained decodes (growing avoid sets) over several maps, and prints how often (a) every node is regular,
(b) delayed nodes are all covered by the chain rule, (c) the model's path and labels equal the sequential ones.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from pathfit import env  # noqa: E402

SO = "/tmp/libsettle_study.so"
subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=gnu99", "-ffp-contract=off", "-fno-fast-math", "-shared", "-o", SO,
                       os.path.join(ROOT, "scripts", "settle", "settle_study.c"), "-lm", "-Wno-unused-function",
                       "-Wno-misleading-indentation"])
L = C.CDLL(SO)
vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
L.orc_ws_create.restype = vp; L.orc_ws_create.argtypes = [i32, i32]
L.settle_v0.restype = i64; L.settle_v0.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i64, vp, vp]
L.ref_v0_labels.restype = i64; L.ref_v0_labels.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i64, vp, vp]
L.settle_v0_T.restype = i64; L.settle_v0_T.argtypes = L.settle_v0.argtypes
L.ref_v1.restype = i64; L.ref_v1.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, i64, vp]


class Study:
    def __init__(self, grid):
        self.occ = np.ascontiguousarray((np.asarray(grid) == 1).astype(np.uint8))
        self.R, self.Cc = self.occ.shape
        self.RC = self.R * self.Cc
        self.ws = L.orc_ws_create(self.R, self.Cc)
        self.out_a = np.zeros(self.RC, np.int32); self.out_b = np.zeros(self.RC, np.int32)
        self.ga = np.zeros(self.RC); self.gb = np.zeros(self.RC)
        self.free = np.flatnonzero(self.occ.reshape(-1) != 1)

    def one(self, s, t, avoid=None, hzero=0):
        st_a = np.zeros(10, np.int64); st_b = np.zeros(6, np.int64)
        av = avoid.ctypes.data if avoid is not None else None
        na = L.settle_v0(self.occ.ctypes.data, self.R, self.Cc, 1, 1, int(s), int(t), av, hzero, self.out_a.ctypes.data, self.RC,
                         st_a.ctypes.data, self.ga.ctypes.data)
        nb = L.ref_v0_labels(self.ws, self.occ.ctypes.data, self.R, self.Cc, 1, 1, int(s), int(t), av, hzero, self.out_b.ctypes.data,
                             self.RC, st_b.ctypes.data, self.gb.ctypes.data)
        same_path = na == nb and np.array_equal(self.out_a[:max(na, 0)], self.out_b[:max(nb, 0)])
        closed = np.isfinite(self.gb)
        same_lab = bool(np.array_equal(self.ga[closed], self.gb[closed]))
        # the T-order certificate on the same input
        st_t = np.zeros(8, np.int64)
        gt = np.zeros(self.RC)
        out_t = np.zeros(self.RC, np.int32)
        nt_ = L.settle_v0_T(self.occ.ctypes.data, self.R, self.Cc, 1, 1, int(s), int(t), av, hzero, out_t.ctypes.data, self.RC,
                            st_t.ctypes.data, gt.ctypes.data)
        t_cert = int(st_t[2]) == 0
        t_ok = True
        if t_cert and int(st_t[6]) == 0:
            t_ok = nt_ == nb and np.array_equal(out_t[:max(nt_, 0)], self.out_b[:max(nb, 0)]) and bool(np.array_equal(gt[closed], self.gb[closed]))
            if nb > 0:
                t_ok = t_ok and int(st_t[0]) + 1 == int(st_b[0])      # expanded nodes + start == the reference's pops
        return dict(n=int(nb), region=int(st_a[0]), delayed=int(st_a[1]), unsafe=int(st_a[2]), expansions=int(st_a[3]),
                    multi=int(st_a[4]), viol=int(st_a[5]), pops=int(st_b[0]), same_path=bool(same_path), same_lab=same_lab,
                    path_delayed=int(st_a[7]), first_delayed_rank=int(st_a[8]), path=self.out_b[:max(nb, 0)].copy(), t_cert=t_cert, t_ok=bool(t_ok), t_why=int(st_t[2]),
                    t_delayed=int(st_t[1]))

    def v1_vs_model(self, s, t):
        st_a = np.zeros(10, np.int64); st_b = np.zeros(6, np.int64)
        na = L.settle_v0(self.occ.ctypes.data, self.R, self.Cc, 1, 1, int(s), int(t), None, 0, self.out_a.ctypes.data, self.RC,
                         st_a.ctypes.data, self.ga.ctypes.data)
        nb = L.ref_v1(self.ws, self.occ.ctypes.data, self.R, self.Cc, 1, 1, int(s), int(t), None, self.out_b.ctypes.data, self.RC,
                      st_b.ctypes.data)
        return na == nb and np.array_equal(self.out_a[:max(na, 0)], self.out_b[:max(nb, 0)]), int(st_b[0]), int(st_a[0])


def summarize(name, rows):
    n = len(rows)
    if not n:
        return
    reg = sum(r["delayed"] == 0 for r in rows)
    safe = sum(r["unsafe"] == 0 for r in rows)
    okp = sum(r["same_path"] for r in rows)
    okl = sum(r["same_lab"] for r in rows)
    bad_safe = sum((r["unsafe"] == 0) and not (r["same_path"] and r["same_lab"]) for r in rows)
    viol = sum(r["viol"] for r in rows)
    pops = sum(r["pops"] for r in rows); exp = sum(r["expansions"] for r in rows)
    dl = sum(r["delayed"] for r in rows)
    tc = sum(r["t_cert"] for r in rows); tbad = sum(r["t_cert"] and not r["t_ok"] for r in rows)
    why = {}
    for r in rows:
        if not r["t_cert"]:
            why[r["t_why"]] = why.get(r["t_why"], 0) + 1
    fb = [r for r in rows if r["delayed"] > 0 and r["region"] > 0]
    if fb:
        fr = np.array([r["first_delayed_rank"] / r["region"] for r in fb])
        wt = np.array([r["pops"] for r in fb], float)
        print(f"{name:28s} searches with delayed nodes: {len(fb)}; share of the region (in key order) before the first one: "
              f"mean {fr.mean():.2f}, pop-weighted {np.average(fr, weights=wt):.2f}, quartiles {np.percentile(fr, [25, 50, 75]).round(2)}", flush=True)
    print(f"{name:28s} T-order certificate: certified {tc}/{n}, certified-but-different {tbad}, give-up reasons {why}", flush=True)
    print(f"{name:28s} cases {n:5d} | all-regular {reg:5d} | certified (no unsafe delayed) {safe:5d} | path== {okp:5d} labels== {okl:5d} | "
          f"certified-but-different {bad_safe} | fixpoint violations {viol} | delayed/search {dl / n:.2f} | expansions/pops {exp / max(pops, 1):.3f}",
          flush=True)


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(7)
    maps = {"G512": env.bench_grid(512), "G256": env.bench_grid(256) if hasattr(env, "bench_grid") else None}
    empty = np.zeros((256, 256), np.int64)
    maps["empty256"] = empty
    rb = env.random_blocks(256, 256, 0.2, seed=5) if hasattr(env, "random_blocks") else None
    if rb is not None:
        maps["blocks256"] = rb
    for name, g in maps.items():
        if g is None:
            continue
        S = Study(g)
        rows = []
        for _ in range(ncase):
            s, t = rng.choice(S.free, 2)
            rows.append(S.one(s, t))
        summarize(name + " pairs", rows)
        rows = []
        for _ in range(max(ncase // 4, 10)):               # chained decode: avoid = cells visited so far
            wps = list(rng.choice(S.free, 5)) + [S.RC - 1 if g.reshape(-1)[S.RC - 1] != 1 else int(S.free[-1])]
            cur = int(S.free[0])
            visited = np.zeros(S.RC, np.uint8); visited[cur] = 1
            for wp in wps:
                r = S.one(cur, int(wp), visited)
                rows.append(r)
                if r["n"] <= 0:
                    break
                visited[r["path"]] = 1
                cur = int(wp)
        summarize(name + " decode-chains", rows)
        rows = []
        for _ in range(max(ncase // 4, 10)):
            s, t = rng.choice(S.free, 2)
            rows.append(S.one(s, t, None, 1))
        summarize(name + " dijkstra", rows)
        same = tot = 0
        for _ in range(max(ncase // 4, 10)):
            s, t = rng.choice(S.free, 2)
            ok, pops, reg = S.v1_vs_model(s, t)
            same += ok; tot += 1
        print(f"{name:28s} MPA._a_star path == closed-set model path: {same}/{tot}", flush=True)


if __name__ == "__main__":
    main()

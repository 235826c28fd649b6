"""Diagnostic (-DPF_TRACE build): how many MPA candidate rebuilds an admissible length bound would prove rejected.
Not part of the product."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), ROOT]
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import dijkstra
from pathfit import _lib
_lib._SO = os.path.join(ROOT, "maaco-path-planing_amd", "lib", "libpathfit_trace.so")
import pathfit
from pathfit import env
from pathfit.dist import Comm, ShardedMPA
import bench
grid = env.bench_grid(512)
R, Cc = grid.shape
occ = (grid == 1)
DR = [0, 0, 1, -1, 1, 1, -1, -1]; DCc = [1, -1, 0, 0, 1, -1, 1, -1]
rows, cols, w = [], [], []
rr, cc = np.meshgrid(np.arange(R), np.arange(Cc), indexing="ij")
for k in range(8):
    nr, nc = rr + DR[k], cc + DCc[k]
    ok = (nr >= 0) & (nr < R) & (nc >= 0) & (nc < Cc) & ~occ
    nrc, ncc = np.clip(nr, 0, R - 1), np.clip(nc, 0, Cc - 1)
    ok &= ~occ[nrc, ncc]
    if k >= 4:
        ok &= ~occ[nrc, cc] & ~occ[rr, ncc]
    rows.append((rr * Cc + cc)[ok]); cols.append((nrc * Cc + ncc)[ok]); w.append(np.full(ok.sum(), 1.0 if k < 4 else 2 ** 0.5))
G = sp.csr_matrix((np.concatenate(w), (np.concatenate(rows), np.concatenate(cols))), shape=(R * Cc, R * Cc))
eng = pathfit.Engine(grid)
eng.L.pf_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
N = 4096
sm = ShardedMPA(Comm(), lambda n: pathfit.MPA(grid, N, 15, engine=eng, seed=0, n_local=n, **bench.MPA_MAIN), N)
m = sm.local
d_s = dijkstra(G, indices=m._s); d_t = dijkstra(G, indices=m._t)
print("L_opt", d_s[m._t], "init fitness", m._stats_host[0])
for it in range(1, 7):
    fit_before = m._stats_host[:, 4].copy()
    cells = m.d_cells.download(); lens = m.d_len.download()
    sm._resort(); gidx, slot = sm._local_view()
    sm.step(it)
    out = np.zeros(12 * 16384, np.uint64)
    eng.L.pf_debug_trace(eng.h, out.ctypes.data)
    t = out[: 4 * 16384].reshape(-1, 4)[: 2 * N].astype(np.int64)
    t2 = out[4 * 16384: 8 * 16384].reshape(-1, 4)[: 2 * N].astype(np.int64)
    t3 = out[8 * 16384:].reshape(-1, 4)[: 2 * N].astype(np.int64)
    pops = t[:, 2]
    # FADs items: a = position in gidx/slot arrays
    fd = np.flatnonzero(t3[N:, 0] > 0)
    node = t3[N + fd, 1]
    lb = d_s[node] + d_t[node]
    pr = lb * (1 - 1e-9) >= fit_before[slot[fd]]
    fp = pops[N + fd]
    print(f"iter {it}: kernel {eng.last_kernel_ms():.1f} ms; FADs searched {len(fd)}, provably rejected {pr.sum()} ({pr.mean():.2f}); pops pruned {fp[pr].sum()} of {fp.sum()}; max pops kept {fp[~pr].max() if (~pr).any() else 0}")
    ph = np.flatnonzero(t3[:N, 0] > 0)
    idx = t3[ph, 0] - 1; inter = t3[ph, 1]; cur = t3[ph, 2]
    pre = np.zeros(len(ph))
    for j, a in enumerate(ph):
        p = cells[slot[a], : idx[j] + 1]
        d = np.abs(np.diff(p // Cc)) + np.abs(np.diff(p % Cc))
        pre[j] = (d == 1).sum() + (d == 2).sum() * 2 ** 0.5
    eu = np.hypot(cur // Cc - inter // Cc, cur % Cc - inter % Cc)
    lbp = pre + eu + d_t[inter]
    prp = lbp * (1 - 1e-9) >= fit_before[slot[ph]]
    pp = pops[ph]
    print(f"        phase rebuilt {len(ph)}, provably rejected {prp.sum()} ({prp.mean():.2f}); pops pruned {pp[prp].sum()} of {pp.sum()}; max pops kept {pp[~prp].max() if (~prp).any() else 0}; fit_before min/med/max {fit_before.min():.1f}/{np.median(fit_before):.1f}/{fit_before.max():.1f}")

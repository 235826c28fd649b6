"""Diagnostic (-DPF_TRACE build): timeline of the fused MPA sweep of the bench workload.  Not part of the product."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), ROOT]
import numpy as np
from pathfit import _lib
_lib._SO = os.path.join(ROOT, "maaco-path-planing_amd", "lib", "libpathfit_trace.so")
import pathfit
from pathfit import env
from pathfit.dist import Comm, ShardedMPA
import bench
grid = env.bench_grid(512)
eng = pathfit.Engine(grid)
eng.L.pf_debug_trace.argtypes = [C.c_void_p, C.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sm = ShardedMPA(Comm(), lambda n: pathfit.MPA(grid, N, 15, engine=eng, seed=0, n_local=n, **bench.MPA_MAIN), N)
for it in range(1, 4):
    sm.step(it)
    out = np.zeros(12 * 16384, np.uint64)
    eng.L.pf_debug_trace(eng.h, out.ctypes.data)
    t = out[: 4 * 16384].reshape(-1, 4)[: 2 * N].astype(np.int64)
    t2 = out[4 * 16384: 8 * 16384].reshape(-1, 4)[: 2 * N].astype(np.int64)
    act = t[:, 1] > 0
    t0 = t[act, 0].min()
    st, en, pops, wave = (t[:, 0] - t0) / 1e5, (t[:, 1] - t0) / 1e5, t[:, 2], t[:, 3]   # 100 MHz clock -> ms
    dur = en - st
    print(f"iter {it}: kernel {eng.last_kernel_ms():.1f} ms; items {act.sum()}; total pops {pops[act].sum()}; end of last item {en[act].max():.1f} ms")
    ph, fd = np.arange(2 * N) < N, np.arange(2 * N) >= N
    for nm, m in (("phase", ph & act), ("fads", fd & act)):
        nz = m & (pops > 0)
        print(f"  {nm}: items with pops {nz.sum()}, pops sum {pops[nz].sum()}, max pops {pops[m].max()}, max dur {dur[m].max():.1f} ms")
    order = np.argsort(-dur * act)[:12]
    for i in order:
        print(f"   item {i} ({'phase' if i < N else 'fads'}): start {st[i]:.1f} end {en[i]:.1f} dur {dur[i]:.1f} ms pops {pops[i]} us/pop {1e3*dur[i]/max(pops[i],1):.2f} wave {wave[i]}")
    for nm, lo in (("phase", 0), ("fads", N)):
        for seg in (0, 1):
            pp, rr = t2[lo:lo + N, 2 * seg], t2[lo:lo + N, 2 * seg + 1]
            ran = rr >= 100
            for res in (0, 1, 2):
                m = ran & (rr == 100 + res)
                if m.any():
                    q = np.percentile(pp[m], [50, 90, 99, 100]).astype(int)
                    print(f"  {nm} A*#{seg + 1} result {res}: n {m.sum()}, pops sum {pp[m].sum()}, p50/p90/p99/max {q.tolist()}")
    # concurrency over time
    print('   duration histogram (ms): ' + ', '.join(f'>{b}: {int((dur[act] > b).sum())}' for b in (2, 5, 8, 10, 12, 14, 16, 20, 24)))
    for T in (2, 5, 8, 10, 12, 14, 16, 20, 24, 28):
        print(f"   t={T} ms: running items {(act & (st <= T) & (en > T)).sum()}", end=";")
    print()

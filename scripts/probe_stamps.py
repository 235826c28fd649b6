"""Diagnostic (-DPF_STAMPS build): where a pop spends its shader cycles.  Not part of the product."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
from pathfit import _lib
_lib._SO = os.path.join(ROOT, "maaco-path-planing_amd", "lib", "libpathfit_stamps.so")
import golden_io as gio
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
e.L.pf_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rnd = np.random.default_rng(1)
free = np.flatnonzero(g.reshape(-1) != 1)
names = ["argmin", "tie-resolve", "bcast+addr+load issue", "rescan (LDS)", "wait load + bcast cur", "relax+push+store", "deckey/ovf/bookkeeping", "#pops with f-ties"]
for n in (32, 1792):
    starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
    ds, dt = e.put(starts), e.put(targets)
    dc, dl, dst = e.buf((n, 8192), np.int32), e.buf(n, np.int32), e.buf(n, np.int32)
    for v in (0, 1):
        out = np.zeros(16, np.uint64)
        e.L.pf_debug_stamps(e.h, out.ctypes.data, 1)
        e.astar_batch(v, ds, dt, n, 8192, dc, dl, dst)
        e.L.pf_debug_stamps(e.h, out.ctypes.data, 1)
        pops = e.counters()["pops"]
        tot = out[:7].sum()
        print(f"n={n} v{v}: {e.last_kernel_ms():.1f} ms, pops {pops}, cycles/pop {tot/pops:.0f}; " +
              "; ".join(f"{names[i]} {out[i]/pops:.0f}" for i in range(7)) + f"; tie pops {out[7]/pops:.3f}")

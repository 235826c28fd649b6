"""Diagnostic (-DPF_STAMPS build): shader-clock time per section of a trip of the sorted-window loop, lone wave."""
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
names = ["refill", "heads+address+load issue", "wait for the loads", "relax+viol+chain", "meta/record stores+pool atomic", "window inserts", "pool stores+tail"]
for n in ((int(sys.argv[1]),) if len(sys.argv) > 1 else (1,)):
    rnd = np.random.default_rng(1)
    free = np.flatnonzero(g.reshape(-1) != 1)
    starts = rnd.choice(free, n).astype(np.int32); targets = rnd.choice(free, n).astype(np.int32)
    starts[0], targets[0] = 0, g.size - 1
    if os.environ.get("PF_SAME"):      # n copies of the corner-to-corner search (shader clocks per trip under load)
        starts[:] = 0; targets[:] = g.size - 1
    for v in (1, 0):
        out = np.zeros(24, np.uint64)
        e.L.pf_debug_stamps(e.h, out.ctypes.data, 1)
        paths, st, cnt = e.astar_host(v, starts, targets, None, path_cap=8192, want_counters=True)
        e.L.pf_debug_stamps(e.h, out.ctypes.data, 1)
        trips = int(out[7]); pops = int(cnt[:, 0].sum())
        print(f"us per trip (kernel time / trips of one search) {1e3 * e.last_kernel_ms() / (trips / n):.3f}")
        print(f"n={n} v{v}: {e.last_kernel_ms():.1f} ms pops {pops} trips {trips} pops/trip {pops / trips:.2f} clocks/trip {out[:7].sum() / trips:.0f}: " +
              "; ".join(f"{names[i]} {out[i] / trips:.0f}" for i in range(7)))
        c = out[8:16].astype(float); er = out[16:].astype(float)
        if er[0] > 0 and not (os.environ.get("PF_TWO_WAVE", "1") != "0" and v == 1):
            print(f"      early refills: {int(er[0])} (every {trips / er[0]:.1f} trips), entries avg {er[1] / er[0]:.1f}, buckets avg {er[3] / er[0]:.1f}, largest bucket avg {er[2] / er[0]:.1f}; "
                  f"clocks per early refill: waiting for the pool entries {er[4] / er[0]:.0f}, sort {er[5] / er[0]:.0f}")
        if os.environ.get("PF_TWO_WAVE", "1") != "0" and v == 1:
            print(f"      two-wave: takes {int(c[0])} (every {trips / max(c[0], 1):.1f} trips), entries per take {c[1] / max(c[0], 1):.1f}, clocks per take {c[2] / max(c[0], 1):.0f}, "
                  f"takes with an empty window {int(c[3])}")
            print(f"      pool wave: draining {er[0] / trips:.0f} clocks per trip ({er[0] / max(er[1], 1):.0f} per entry, {er[1] / trips:.1f} entries per trip); background refills {int(er[3])} "
                  f"({er[2] / max(er[3], 1):.0f} clocks each, {er[2] / trips:.0f} per trip); serving {er[4] / max(c[0], 1):.0f} clocks per take, {int(er[5])} refills at serve time")
        print(f"      refills: front small {int(c[0])} (avg {c[1] / max(c[0], 1):.1f}), front big {int(c[2])} (avg size {c[3] / max(c[2], 1):.1f}), "
              f"regular {int(c[6])} (avg {c[7] / max(c[6], 1):.1f}), regular big {int(c[4])} (avg size {c[5] / max(c[4], 1):.1f})")

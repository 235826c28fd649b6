"""Diagnostic: ONE MPA rebuild (idx 3 of the initial path) alone on the chip through k_mpa_phase, to compare the
connector's speed inside the MPA kernels with the batch kernel (scripts/probe_phase_like.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import golden_io as gio
from pathfit.engine import Engine, score_params
from pathfit._lib import MpaParams
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
s, t = 0, g.size - 1
sp = score_params(1, True, 0.1, 0.8, 1.8, 100.0)
e.mpa_setup(MpaParams(0.5, 2.0, 0.6966, 0.2, 1, s, t, 1, 1), sp)
paths, st = e.astar_host(1, [s], [t], None, path_cap=8192)
p0 = paths[0]
cap = 8256
for idx in (3, 660):
    pop = np.zeros((1, cap), np.int32); pop[0, :len(p0)] = p0
    dpop, dlen, dstats, del_ = e.put(pop), e.put(np.array([len(p0)], np.int32)), e.put(np.zeros((1, 5))), e.put(p0)
    oc, ol, os_, ost = e.buf((1, cap), np.int32), e.buf(1, np.int32), e.buf((1, 5), np.float64), e.buf(1, np.int32)
    d_idx, d_lv, d_sc, d_ag = e.put(np.array([idx], np.int32)), e.put(np.array([0], np.int32)), e.put(np.array([0.5])), e.put(np.array([0], np.int32))
    for rep in range(2):
        e._ck(e.L.pf_mpa_rebuild_batch(e.h, 1, 0, 1, cap, dpop.ptr, dlen.ptr, dstats.ptr, del_.ptr, len(p0), d_idx.ptr, d_lv.ptr, d_sc.ptr, d_ag.ptr,
                                       oc.ptr, ol.ptr, os_.ptr, ost.ptr))
    c = e.counters()
    print(f"idx {idx}: {e.last_kernel_ms():.2f} ms pops {c['pops']} us/pop {1e3 * e.last_kernel_ms() / max(c['pops'], 1):.3f} status {ost.download()[0]} len {ol.download()[0]}")

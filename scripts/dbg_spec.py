import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from pathfit import _lib
_lib._SO = os.path.join(ROOT, "maaco-path-planing_amd", "lib", "libpathfit_dbg.so")
import golden_io as gio
from pathfit.engine import Engine
z = gio.load("astar_cases")
i = int(sys.argv[1]) if len(sys.argv) > 1 else 120
names = [str(s) for s in z["grid_names"]]
g, s0, t0 = gio.grid(names[int(z["grid_id"][i])])
variant = int(z["variant"][i])
start, target = int(z["start"][i]), int(z["target"][i])
avoid = gio.csr_get(z["avoid_off"], z["avoid"], i) if z["has_avoid"][i] else None
R, C = g.shape
print("case", i, names[int(z["grid_id"][i])], "variant", variant, "start", divmod(start, C), "target", divmod(target, C), "avoid", None if avoid is None else len(avoid), "want pops", int(z["pops"][i]))
# sequential reference order (variant 0: closed set + decrease-key)
occ = g == 1
MOVES = [(0,1),(0,-1),(1,0),(-1,0),(1,1),(1,-1),(-1,1),(-1,-1)]
av = set() if avoid is None else set(int(x) for x in avoid)
def nb(r, c):
    for dr, dc in MOVES:
        nr, nc = r+dr, c+dc
        if nr < 0 or nr >= R or nc < 0 or nc >= C or occ[nr, nc]: continue
        if dr and dc and (occ[nr, c] or occ[r, nc]): continue
        yield nr, nc, (1.0 if dr == 0 or dc == 0 else math.sqrt(2.0))
tr, tc = divmod(target, C)
h = lambda r, c: math.sqrt((r-tr)**2 + (c-tc)**2)
sr, sc = divmod(start, C)
openl = {(sr, sc): (h(sr, sc), 0.0)}
gs = {(sr, sc): 0.0}; closed = set(); k = 0
while openl and variant == 0:
    cell = min(openl, key=lambda c: (openl[c][0], openl[c][1], c))
    f, gg = openl.pop(cell); k += 1
    print(f"  seq pop {k}: f={f:.6f} g={gg:.4f} rc={cell}")
    if cell == (tr, tc): break
    closed.add(cell)
    for nr, nc, cost in nb(*cell):
        n = (nr, nc)
        if n in closed: continue
        if (nr*C+nc) in av and n != (sr, sc) and n != (tr, tc): continue
        t = gg + cost
        if n not in gs or t < gs[n]:
            gs[n] = t; openl[n] = (t + h(nr, nc), t)
e = Engine(g)
paths, st, cnt = e.astar_host(variant, [start], [target], [avoid] if avoid is not None else None, want_counters=True)
print("gpu pops", cnt[0], "status", st[0])

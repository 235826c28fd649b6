import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "maaco-path-planing_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
from pathfit import _lib
_lib._SO = os.path.join(ROOT, "maaco-path-planing_amd", "lib", "libpathfit_dbg.so")
import golden_io as gio
from pathfit.engine import Engine
g = gio.upsample(gio.grid("g256")[0], 2)
e = Engine(g)
rnd = np.random.default_rng(5)
free = np.flatnonzero(g.reshape(-1) != 1)
n = 48
starts = rnd.choice(free, n); targets = rnd.choice(free, n)
avoid = [rnd.choice(free, 200) if i % 2 else None for i in range(n)]
i = int(sys.argv[1]); v = int(sys.argv[2])
paths, st, cnt = e.astar_host(v, starts[i:i+1], targets[i:i+1], [avoid[i]], path_cap=8192, want_counters=True)
print("status", st, "cnt", cnt, "start", divmod(int(starts[i]), 512), "target", divmod(int(targets[i]), 512))

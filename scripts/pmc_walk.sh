#!/bin/bash
# PMC passes over one workload's dominant kernel: bash scripts/pmc_walk.sh <outdir> <workload> "<counters pass 1>" "<pass 2>" ...
OUT=$(realpath "$1"); W=$2; shift 2; mkdir -p "$OUT"; ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
i=0
for C in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/p$i" -- python3 "$ROOT/bench.py" --workload $W --no-cpu --no-extra --steps 4 --warmup 1 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/p$i.log"; exit 1; }
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("k_maaco_walk", "k_mpa_search", "k_decode_batch", "k_astar_batch")):
            a = acc[(r["Kernel_Name"][:40], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print(f"{k:42s} {c:28s} per-launch {v / n:16.1f}  (n={n})")
PY
  rm -rf "$OUT/p$i"
done

#!/bin/bash
# kernel time of the pheromone update under rocprofv3 for each library given (PF_LIB A/B); run on the GPU box from the repo root:
#   bash scripts/prof_tau.sh maaco512 lib1.so [lib2.so ...]
W=$1; shift
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
for L in "$@"; do
  export PF_LIB=$ROOT/$L
  rm -rf /tmp/p_tau
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_tau -- python3 $ROOT/bench.py --workload $W --no-cpu --no-extra --steps 10 --warmup 3 > /tmp/p_tau.log 2>&1 || { echo "$L failed"; tail -5 /tmp/p_tau.log; exit 1; }
  echo "== $L: $(grep '^{' /tmp/p_tau.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  f=$(find /tmp/p_tau -name "*kernel_stats.csv" | head -1)
  grep "k_tau_update\|k_maaco_walk" $f | sed "s/(.*)\"/\"/" | cut -d, -f1-4
  python3 - $(find /tmp/p_tau -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_tau_update" in r["Kernel_Name"]]
print("   k_tau_update per call (us):", " ".join(str((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) // 1000) for r in rows))
PY
done

#!/bin/bash
# One round of profiling runs on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh gpurun_out/prof_r04 [workloads...]
# per workload: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, as
# MI355X_MICROARCH.md prescribes: the two do not fit one pass) and the unprofiled bench line; scripts/make_traffic.py turns
# them into profiles/<round>_traffic.json + profiles/<round>_<workload>_kernel_stats.csv (PF_ROUND, default r04) as the LAST STEP OF THIS SCRIPT, so the
# traffic figure bench.py quotes is always captured with the kernels it describes.  The program goes straight after `--`
# (python3 itself: no env / bash -c hop, the profiler's preload has already initialised the GPU).
set -o pipefail
OUT=$(realpath "$1"); shift; mkdir -p "$OUT"
WL=${@:-mpa512 maaco512 maaco128 maaco1024 ga512 pso512 astar1024}
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for W in $WL; do
  # mpa512 runs the driver's protocol: the traffic figure is keyed to this kernel time
  case $W in mpa512) ST="--steps 20 --warmup 5";; ga512|astar1024|pso512) ST="--steps 2 --warmup 1";; maaco128) ST="--steps 100 --warmup 5";; *) ST="--steps 10 --warmup 2";; esac
  timeout -k 10 200 python3 "$ROOT/bench.py" --workload $W --no-cpu --no-extra --no-copy-probe $ST > "$OUT/${W}_bench.json" 2> "$OUT/${W}_bench.err" || { echo "$W bench failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${W}_stats" -- python3 "$ROOT/bench.py" --workload $W --no-cpu --no-extra --no-copy-probe $ST > "$OUT/${W}_stats.log" 2>&1 || { echo "$W stats failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${W}_fetch" -- python3 "$ROOT/bench.py" --workload $W --no-cpu --no-extra --no-copy-probe $ST > "$OUT/${W}_fetch.log" 2>&1 || { echo "$W fetch failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${W}_write" -- python3 "$ROOT/bench.py" --workload $W --no-cpu --no-extra --no-copy-probe $ST > "$OUT/${W}_write.log" 2>&1 || { echo "$W write failed"; exit 1; }
  echo "$W done"
done
# keep the merge-back small: the per-dispatch trace is reduced to the workload's dominant kernel (make_traffic.py averages the
# dispatches of the timed region only), everything else but the summaries goes
for W in $WL; do
  for f in $(find "$OUT/${W}_stats" -name '*kernel_trace.csv'); do
    python3 - "$f" "$OUT/${W}_dominant_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_mpa_search", "k_maaco_walk", "k_decode_batch", "k_astar_batch<"))]
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
for r in keep:
    w.writerow([r["Kernel_Name"], r["Start_Timestamp"], r["End_Timestamp"]])
PY
  done
done
find "$OUT" -type f \( -name '*.db' -o -name '*kernel_trace.csv' -o -name '*agent_info.csv' \) -delete
cd "$ROOT"
python3 scripts/make_traffic.py "$OUT" > "$OUT/traffic_summary.log" 2>&1 || true
cp profiles/${PF_ROUND:-r04}_traffic.json profiles/${PF_ROUND:-r04}_*_kernel_stats.csv "$OUT/" 2>/dev/null || true
tail -60 "$OUT/traffic_summary.log"

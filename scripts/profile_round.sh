#!/bin/bash
# One round of profiling runs on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh gpurun_out/prof_r01k
# per workload: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, as
# MI355X_MICROARCH.md prescribes) and the unprofiled bench line; scripts/make_traffic.py turns them into
# profiles/r01_traffic.json + profiles/r01_<workload>_kernel_stats.csv.
set -e -o pipefail
OUT=$(realpath "$1"); mkdir -p "$OUT"
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
for W in mpa512 maaco512; do
  timeout -k 10 200 python3 "$ROOT/bench.py" --workload $W --no-cpu > "$OUT/${W}_bench.json" 2> "$OUT/${W}_bench.err"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${W}_stats" -- python3 "$ROOT/bench.py" --workload $W --no-cpu > "$OUT/${W}_stats.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${W}_fetch" -- python3 "$ROOT/bench.py" --workload $W --no-cpu > "$OUT/${W}_fetch.log" 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${W}_write" -- python3 "$ROOT/bench.py" --workload $W --no-cpu > "$OUT/${W}_write.log" 2>&1
  echo "$W done"
done
# keep the merge-back small: drop everything but the CSVs the summary needs
find "$OUT" -type f \( -name '*.db' -o -name '*kernel_trace.csv' -o -name '*agent_info.csv' \) -delete
cd "$ROOT"
python3 scripts/make_traffic.py "$OUT" > "$OUT/traffic_summary.log" 2>&1 || true
cp profiles/r01_traffic.json profiles/r01_mpa512_kernel_stats.csv profiles/r01_maaco512_kernel_stats.csv "$OUT/" 2>/dev/null || true
tail -40 "$OUT/traffic_summary.log"

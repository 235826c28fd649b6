# A/B of the parallel engine policy on the GPU box (repo root): tail thresholds (per mille of the search slots) on ga512 / pso512, and always / never on astar1024.
# r03, one box: tail 400 / 600 / 800 / 1000 -> ga512 19.6 / 20.6 / 20.5 / 17.6 k evals/s; astar1024 always 34.6 k, never 45.8 k solves/s.
for T in 400 600 800 1000; do
  for w in ga512 pso512; do
    PF_SETTLE_TAIL=$T timeout -k 10 200 python bench.py --workload $w --no-cpu --no-extra --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail $T', '$w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
PF_SETTLE=1 timeout -k 10 200 python bench.py --workload ga512 --no-cpu --no-extra --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle always ga512', d['value'], d['ms_per_step'])"
PF_SETTLE=1 timeout -k 10 200 python bench.py --workload astar1024 --no-cpu --no-extra --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle always astar1024', d['value'], d['ms_per_step'])"
PF_SETTLE=0 timeout -k 10 200 python bench.py --workload astar1024 --no-cpu --no-extra --steps 2 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle never astar1024', d['value'], d['ms_per_step'])"

for st in -1 1; do
  for w in ga512 pso512 astar1024; do
  PF_SETTLE=$st timeout -k 10 300 python bench.py --workload $w --no-extra --no-cpu --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('PF_SETTLE=$st', '$w', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done

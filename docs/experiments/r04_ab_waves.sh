# resident search waves per CU (PF_WAVES_PER_CU caps what the occupancy query allows)
for wl in astar1024 mpa512 ga512; do
for v in 8 12 16; do
  PF_WAVES_PER_CU=$v python bench.py --workload $wl --steps ${STEPS:-3} --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$wl waves/CU', $v, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done

# ants per wavefront of k_maaco_walk8 (8 lanes each): 8 = one wave per SIMD at 8192 ants, 4 = two, 2 = four
for wl in maaco1024 maaco512; do
for v in 8 4 2; do
  PF_MAACO_GROUPS=$v python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$wl ants/wave', $v, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done

for v in 0 96 64 128; do
  PF_MPA_LONG_CUS=$v python bench.py --steps 20 --warmup 5 --no-cpu --no-extra 2>gpurun_out/r4_ab_long_$v.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('long_cus', $v, d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['sweep_ms_by_iteration'][:6])"
done

# interleaved A/B of one environment switch on one box: scripts/ab_env.sh VAR "v0 v1" "workload steps" ...
VAR=$1; VALS=$2; shift 2
for rep in 1 2; do for wl in "$@"; do W=${wl% *}; ST=${wl#* }; for v in $VALS; do
  env $VAR=$v python bench.py --workload $W --steps $ST --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$W', '$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done; done; done

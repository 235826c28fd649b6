# A/B of two builds of the library on one box, interleaved: scripts/ab_libs.sh libA.so libB.so
A=$1; B=$2
run() { PF_LIB=$1 python bench.py --workload $2 --steps $3 --warmup 1 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$2', '$(basename $1)', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
for wl in "astar1024 3" "ga512 3" "pso512 3" "mpa512 20"; do set -- $wl; run $A $1 $2; run $B $1 $2; done
done

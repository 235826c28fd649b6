# A/B of two builds of the library on one box, interleaved: scripts/ab_libs.sh libA.so libB.so "workload steps" ...
A=$1; B=$2; shift 2
run() { PF_LIB=$1 python bench.py --workload $2 --steps $3 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$2', '$(basename $1)', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do for wl in "$@"; do W=${wl% *}; ST=${wl#* }; run $A $W $ST; run $B $W $ST; done; done

# interleaved A/B on one box: old library (libA) against the new one (libB) with the load-ahead form of k_maaco_walk8 never / always / by occupancy
run() { env PF_LIB=$1 PF_MAACO_TOUCH=$2 python bench.py --workload $3 --steps $4 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$3', '$(basename $1)', 'ahead=$2', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
L=maaco-path-planing_amd/lib/ab
for rep in 1 2; do for wl in "maaco512 20" "maaco1024 20"; do W=${wl% *}; ST=${wl#* }; run $L/libA.so -1 $W $ST; run $L/libB.so 0 $W $ST; run $L/libB.so 1 $W $ST; run $L/libB.so -1 $W $ST; done; done

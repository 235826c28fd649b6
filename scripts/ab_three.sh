# interleaved A/B/C of three builds of the library on one box: scripts/ab_three.sh ["workload steps" ...]
run() { PF_LIB=$1 python bench.py --workload $2 --steps $3 --warmup 2 --no-cpu --no-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$2', '$(basename $1)', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
[ $# -eq 0 ] && set -- "mpa512 20" "ga512 3" "astar1024 3"
for rep in 1 2; do for wl in "$@"; do W=${wl% *}; ST=${wl#* }; for L in A B C; do run maaco-path-planing_amd/lib/ab/lib$L.so $W $ST; done; done; done

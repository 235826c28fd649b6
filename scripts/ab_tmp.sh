for lib in "" maaco-path-planing_amd/lib/libpathfit_q128.so maaco-path-planing_amd/lib/libpathfit_q256.so maaco-path-planing_amd/lib/libpathfit_prev.so; do
  echo "lib=$lib"
  PF_LIB=$lib timeout -k 10 200 python bench.py --no-extra --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['sweep_ms_by_iteration'][:4])"
  PF_LIB=$lib timeout -k 10 200 python bench.py --workload ga512 --no-extra --no-cpu --steps 2 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('ga512', d['value'], d['ms_per_step'])"
done
for lib in maaco-path-planing_amd/lib/libpathfit_q128.so maaco-path-planing_amd/lib/libpathfit_q256.so; do
PF_LIB=$lib timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
done

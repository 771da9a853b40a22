cd $GRAFT_REPO_ROOT
for load in 0.6 0.7 0.6 0.7; do
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --e2e-pairs 0 --steps 10 --warmup 3 --load $load 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('load $load', d['config']['table_buckets'], round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items()})"
done

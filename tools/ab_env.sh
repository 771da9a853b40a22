# A/B of an environment switch on ONE box: bench.py without and with "$1" (NAME=value), alternating.  usage: bash tools/ab_env.sh PG_COUNT_GRID=buckets
set -e
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for which in default "$1"; do
    if [ "$which" = default ]; then pre=""; else pre="$1"; fi
    env $pre timeout -k 10 200 python3 bench.py --no-cpu-baseline --e2e-pairs 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$which', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items()})"
  done
done

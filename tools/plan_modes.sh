#!/bin/bash
# ON the GPU box: the bench in its plan modes
cd /tmp && export TMPDIR=/tmp
for m in ahead in-step once; do
  python3 $GRAFT_REPO_ROOT/bench.py --plan $m --no-cpu-baseline --e2e-pairs 0 --steps 6 --warmup 2 2>/dev/null | grep "^{" | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$m', round(j['ms_per_step'],2), j['kernel_ms'], round(j['value']/1e6,1))"
done
PG_PLAN_WHERE=second-pass python3 $GRAFT_REPO_ROOT/bench.py --plan ahead --no-cpu-baseline --e2e-pairs 0 --steps 6 --warmup 2 2>/dev/null | grep "^{" | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('beside second pass', round(j['ms_per_step'],2), j['kernel_ms'], round(j['value']/1e6,1))"

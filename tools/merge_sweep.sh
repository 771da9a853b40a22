#!/bin/bash
# ON the GPU box: the merged lookups with other step / stage sizes, and the count phase alone (A/B on one box)
cd $GRAFT_REPO_ROOT
cp pangaea_amd/libpangaea_feat.so /tmp/lib_plain.so
run() { PG_MINI_MERGE=$1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --e2e-pairs 0 --steps 8 --warmup 2 2>/dev/null | python3 -c "
import json,sys;d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]);print('$2', 'merge=$1', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items()})"; }
run 0 default; run 1 default
for flags in "-DPG_MERGE_NS=8" "-DPG_MERGE_NS=16" "-DPG_MERGE_NS=12 -DPG_MERGE_WPL=10" "-DPG_DIAG_COUNT=2"; do
  (cd pangaea_amd/csrc && touch mini.hip && make -s KFLAGS="$flags" all > /dev/null 2>&1)
  run 1 "$flags"
  if [ "$flags" = "-DPG_DIAG_COUNT=2" ]; then run 0 "$flags"; fi
done
cp /tmp/lib_plain.so pangaea_amd/libpangaea_feat.so

# bench.py with the in-tree library and with each library given, two rounds, on ONE box (diagnostic builds: timings only)
# usage: bash tools/ab_libs.sh a.so b.so ...
set -e
cd $GRAFT_REPO_ROOT
cp pangaea_amd/libpangaea_feat.so /tmp/lib_new.so
for round in 1 2; do
  for which in new "$@"; do
    if [ $which = new ]; then cp /tmp/lib_new.so pangaea_amd/libpangaea_feat.so; else cp $which pangaea_amd/libpangaea_feat.so; fi
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --e2e-pairs 0 --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$which', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items()})" || echo "$which failed"
  done
done
cp /tmp/lib_new.so pangaea_amd/libpangaea_feat.so

# bench.py with the in-tree library and with each variant library given, two rounds, on ONE box (variants: timings only, their
# results may be wrong).  Variants are selected with PANGAEA_LIB=<path> PANGAEA_ALLOW_VARIANT=1 -- nothing is ever copied over
# the product library.  Build them with `make -C pangaea_amd/csrc variant NAME=x KFLAGS=...` (-> build/libpangaea_feat_x.so).
# usage: bash tools/ab_libs.sh [--] a.so b.so ...      (BENCH_ARGS="..." adds bench.py arguments)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for which in product "$@"; do
    if [ $which = product ]; then sel=""; else sel="PANGAEA_LIB=$which PANGAEA_ALLOW_VARIANT=1"; fi
    env $sel timeout -k 10 200 python3 bench.py --no-cpu-baseline --e2e-pairs 0 --steps 10 --warmup 3 $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$which', round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items()})" || echo "$which failed"
  done
done

# rocprofv3 kernel averages of bench.py for the in-tree library and each variant library given (variants: timings only).
# Variants are selected with PANGAEA_LIB=<path> PANGAEA_ALLOW_VARIANT=1 (exported: rocprofv3 must start python itself, not env).
# usage: bash tools/kernel_time_libs.sh pattern a.so b.so ...
pat=$1; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for which in product "$@"; do
  if [ $which = product ]; then unset PANGAEA_LIB PANGAEA_ALLOW_VARIANT; else export PANGAEA_LIB=$(realpath $which) PANGAEA_ALLOW_VARIANT=1; fi
  rm -rf /tmp/kt_prof
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 4 --warmup 1 $BENCH_ARGS > /dev/null 2>&1)
  python3 - "$which" "$pat" <<'PY'
import csv, glob, sys
which, pat = sys.argv[1], sys.argv[2]
for f in glob.glob('/tmp/kt_prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(p in r['Name'] for p in pat.split(',')):
            print(which, r['Name'][:70].replace('(anonymous namespace)::',''), r['Calls'], round(float(r['AverageNs']) / 1e6, 3))
PY
done
unset PANGAEA_LIB PANGAEA_ALLOW_VARIANT

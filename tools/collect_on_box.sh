#!/bin/bash
# run ON the GPU box (through gpurun): bench line, rocprofv3 kernel stats of the same command, the two PMC passes;
# the bulky rocprofv3 output stays in /tmp, the condensed summaries land in profiles/ and are copied to gpurun_out/
set -eo pipefail
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal_f -- python3 $ROOT/tools/fetch_calibration.py run > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/cal_w -- python3 $ROOT/tools/fetch_calibration.py run > /dev/null 2>&1
cd $ROOT
python3 tools/collect_profiles.py $TAG /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
python3 tools/fetch_calibration.py report /tmp/cal_f /tmp/cal_w > profiles/${TAG}_fetch_calibration.txt
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_fetch_calibration.txt profiles/traffic.json gpurun_out/profiles_$TAG/
grep "^{" gpurun_out/${TAG}_bench.json | tail -1

#!/bin/bash
# run ON the GPU box (through gpurun): bench line, rocprofv3 kernel stats of the same command, the two PMC passes, and one
# rank's share of an 8-rank step rehearsed on this GPU; the bulky rocprofv3 output stays in /tmp, the condensed summaries land in
# profiles/ and are copied to gpurun_out/
set -eo pipefail
TAG=${1:-r03a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $ROOT/gpurun_out/${TAG}_bench.json 2> $ROOT/gpurun_out/${TAG}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -- python3 $ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/cal_f -- python3 $ROOT/tools/fetch_calibration.py run > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/cal_w -- python3 $ROOT/tools/fetch_calibration.py run > /dev/null 2>&1
# one rank's share of an 8-rank step (the N > 1 code path over a one-rank RCCL group): bench line + kernel statistics
python3 $ROOT/bench.py --rehearse-dist 8 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_bench_rehearse_n8_path.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_reh -- python3 $ROOT/bench.py --rehearse-dist 8 --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
cd $ROOT
python3 tools/collect_profiles.py rehearse $TAG /tmp/prof_reh
grep "^{" gpurun_out/${TAG}_bench_rehearse_n8_path.json | tail -1 > profiles/${TAG}_bench_rehearse_n8_path.json
grep "^{" gpurun_out/${TAG}_bench.json | tail -1 > profiles/${TAG}_bench.json
grep "^{" gpurun_out/${TAG}_bench_under_rocprof.json | tail -1 > profiles/${TAG}_bench_under_rocprof.json
python3 tools/collect_profiles.py $TAG /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
python3 tools/fetch_calibration.py report /tmp/cal_f /tmp/cal_w > profiles/${TAG}_fetch_calibration.txt
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* profiles/traffic.json gpurun_out/profiles_$TAG/
cat profiles/${TAG}_bench.json

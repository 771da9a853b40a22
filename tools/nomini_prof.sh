cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nm -- python3 $GRAFT_REPO_ROOT/bench.py --no-mini --no-cpu-baseline --e2e-pairs 0 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/b_nomini.json 2>/dev/null
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_nm | grep -E "scatter|bucket_|row_hist|features"
grep "^{" $GRAFT_REPO_ROOT/gpurun_out/b_nomini.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['kernel_ms'])"

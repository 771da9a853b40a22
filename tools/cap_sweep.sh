cd /tmp && export TMPDIR=/tmp
for cap in ${@:-12 8 6 4}; do
  export PG_MINI_CAP=$cap; rm -rf /tmp/prof_cap$cap
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cap$cap -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --e2e-pairs 0 --steps 4 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/b_cap$cap.json 2>/dev/null
  echo "cap $cap" | tee -a $GRAFT_REPO_ROOT/gpurun_out/cap_sweep.txt
  python3 $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/prof_cap$cap | grep -E "mini_|scatter_rec|row_hist" | tee -a $GRAFT_REPO_ROOT/gpurun_out/cap_sweep.txt
done

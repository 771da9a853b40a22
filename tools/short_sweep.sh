#!/bin/bash
# ON the GPU box: rebuild with other short/long class boundaries and time the count kernel (tuning aid)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for m in "$@"; do
  make -C $ROOT/pangaea_amd/csrc clean > /dev/null
  make -C $ROOT/pangaea_amd/csrc -j8 KFLAGS=-DPG_SHORT_MAX=$m > /dev/null 2>&1
  echo "== SHORT_MAX=$m"
  (cd $ROOT && timeout -k 10 200 python3 -m pytest tests/test_mini_gpu.py -x -q 2>&1 | tail -1 | grep -q " passed") || { echo FAILED; exit 1; }
  rm -f $ROOT/gpurun_out/cap_sweep.txt
  timeout -k 10 300 bash $ROOT/tools/cap_sweep.sh 9 | grep "mini_count\|mini_scatter2"
done

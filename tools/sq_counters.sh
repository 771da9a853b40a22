#!/bin/bash
# ON the GPU box: SQ counters of one bench step per kernel (two passes: the counters do not all fit one)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq1 /tmp/sq2 /tmp/sq3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --plan once --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/sq1 | grep -v "scan_kernel\|total\|round_rows\|group_caps"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --plan once --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/sq2 | grep -v "scan_kernel\|total\|round_rows\|group_caps"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d /tmp/sq3 -- python3 $GRAFT_REPO_ROOT/bench.py --plan once --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/sq3 | grep -v "scan_kernel\|total\|round_rows\|group_caps"

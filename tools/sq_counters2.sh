#!/bin/bash
# ON the GPU box: LDS / instruction-fetch / vector-memory-path counters of one bench step per kernel (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/sqx
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/sqx -- python3 $GRAFT_REPO_ROOT/bench.py --plan once --no-cpu-baseline --e2e-pairs 0 --steps 1 --warmup 0 > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/sqx | grep -v "scan_kernel\|total\|round_rows\|group_caps\|sketch\|normalize"
}
run SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_LDS
run SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_INSTS_BRANCH
run SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES
run SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM
run TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum

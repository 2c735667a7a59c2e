#!/bin/bash
# SQ / TCC counters of a prefill kernel on 4 x 4096 tokens, 32q/8kv (the diagnostic harness binary
# tools/prefill_p0 = tools/prefill_diag.hip built without stamps), one rocprofv3 --pmc run per
# counter group. usage: tools/pmc_prefill.sh <kernel_hint> <outdir>
#   kernel_hint: an LV_PREFILL_KERNEL_* value (3 = phase-locked 8-wave kernel, 5 = 64 rows per wave)
# then: python tools/pmc_prefill_summary.py <outdir> <tag>
HINT=$1
OUT=$2
BIN=$PWD/tools/prefill_p0
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  local name=$1; shift
  timeout -s KILL 90 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- $BIN 4096 $HINT > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; }
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
run b SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
run c SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_SALU SQ_IFETCH SQ_INST_LEVEL_LDS || exit 1
run d TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum || exit 1
run e SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || echo "pass e skipped"
echo done $HINT

#!/bin/bash
# Collects SQ counters for the prefill kernel (diagnostic harness binary) in separate passes.
# usage: tools/pmc_prefill.sh <kernel_hint> <outdir>   (LV_PREFILL_KERNEL_* value: 3 = phase-locked, 5 = 64 rows per wave)
set -e
HINT=$1
OUT=$2
BIN=$PWD/tools/prefill_p0
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/a -o a --output-format csv -- $BIN 4096 $HINT > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/b -o b --output-format csv -- $BIN 4096 $HINT > /dev/null 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_SALU SQ_IFETCH SQ_INST_LEVEL_LDS -d $OUT/c -o c --output-format csv -- $BIN 4096 $HINT > /dev/null 2>&1
echo done $1

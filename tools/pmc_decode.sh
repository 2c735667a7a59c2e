#!/bin/bash
# HBM traffic of the bench's decode kernel from PMC counters: two separate passes (FETCH_SIZE,
# WRITE_SIZE), each a counter-collection run of bench.py with two steps and its config-4 (32k context) leg. Output under gpurun_out/.
set -e
R=$PWD
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 400 rocprofv3 --pmc $c --kernel-include-regex "decode_(attn|mfma)_kernel" --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --aux config4 > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv"

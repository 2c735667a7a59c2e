#!/bin/bash
# Kernel-trace summaries of the bench, the prefill sweep and the engine run (rocprofv3), written
# under gpurun_out/; the *_kernel_stats.csv files are the ones copied into profiles/.
set -e
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --no-aux > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prefill -o prefill -- python3 $R/tools/kbench.py --what prefill > $R/gpurun_out/prof_prefill.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -o e2e -- python3 $R/tools/e2e_bench.py --model 8b --batch 256 --prompt-len 1024 --new-tokens 33 --modes graph > $R/gpurun_out/prof_e2e.log 2>&1
cd $R
find gpurun_out/prof_bench gpurun_out/prof_prefill gpurun_out/prof_e2e -name "*kernel_stats.csv" | head
find gpurun_out -name "*kernel_trace.csv" -delete

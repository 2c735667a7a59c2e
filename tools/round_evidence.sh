#!/bin/bash
# One GPU call's worth of recordings: the config-4 shard shapes under graph replay, the one-sequence
# split-count ablation, the prefill counter passes and the decode traffic passes. Every step has its
# own time limit and the script stops at the first failure.
set -e
R=$PWD
mkdir -p gpurun_out
timeout -k 10 240 python3 -u tools/kbench.py --what decode_c4_tp > gpurun_out/decode_c4_tp.log 2>&1
echo "c4_tp done"
bash tools/decode_b1_ablation.sh > /dev/null
echo "b1 ablation done"
bash tools/pmc_prefill.sh 3 $R/gpurun_out/pmc_pp
echo "prefill counters done"
bash tools/pmc_decode.sh
echo "decode counters done"

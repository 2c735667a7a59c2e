#!/bin/bash
# Samples the GPU's power and clocks (rocm-smi, read-only) while one kernel runs back to back for
# ~6 s: the prefill kernel, the decode kernel, and the bare MFMA burn of tools/mfma_peak.
# Output: gpurun_out/power_probe.log
OUT=gpurun_out/power_probe.log
: > $OUT
sample() {  # label
  for i in 1 2 3 4; do
    sleep 1
    echo "--- $1 sample $i" >> $OUT
    timeout 10 rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk|socclk" >> $OUT
  done
}
echo "=== idle" >> $OUT; sample idle
timeout -k 10 60 python tools/kbench.py --what prefill_burn >> $OUT 2>&1 &
P=$!; sleep 1.5; sample prefill; wait $P
timeout -k 10 60 python tools/kbench.py --what decode_burn >> $OUT 2>&1 &
P=$!; sleep 1.5; sample decode; wait $P
timeout -k 10 60 ./tools/mfma_peak 2 1 >> $OUT 2>&1 &
P=$!; sample mfma_burn; wait $P

#!/bin/bash
# Stamped ablation runs of the phase-locked prefill kernel (cfg 2). Build the variants on the CPU
# box first, one binary per mask (bits: see LV_ABL in light_vllm_amd/csrc/prefill_attn.hip):
#   for a in 0 512 640 8832 704; do hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLV_STAMPS \
#     -DLV_ABL=$a -Iinclude -Ilight_vllm_amd/csrc tools/prefill_diag.hip -o tools/prefill_abl$a; done
# then run this script on the GPU: per-cluster cycles per (wave, tile) for every variant.
HINT=${LV_PREFILL_HINT:-3}   # LV_PREFILL_KERNEL_PP
for a in ${ABL_MASKS:-0 512 640 8832 704}; do
  echo "== ABL $a"
  timeout -k 10 60 tools/prefill_abl$a 4096 $HINT | grep -E "launches|C_QK|C_PV|L_K:|L_V|total" || exit 1
done

#!/bin/bash
# Builds diagnostic copies of the library with pieces of the vector-ALU decode kernel's arithmetic
# compiled out (LV_DECODE_ABL, see csrc/decode_attn.hip) into tools/abl/, for
#   python tools/kbench.py --what decode --lib tools/abl/liblvattn_abl<mask>.so
# Results of those copies are wrong by design; only their launch times mean something.
set -e
cd "$(dirname "$0")/../light_vllm_amd/csrc"
mkdir -p ../../tools/abl
for mask in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -w -DLV_DECODE_ABL=$mask -shared \
    api.hip store_kvcache.hip fused_ops.hip decode_attn.hip prefill_attn.hip comm.hip -o ../../tools/abl/liblvattn_abl$mask.so &
done
wait

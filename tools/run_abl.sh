export LV_PREFILL_CFG=2
for a in 640 8832 704; do echo "== ABL $a"; timeout -k 10 60 tools/prefill_abl$a 4096 | grep -E "launches|C_QK|C_PV|L_K:|total" || exit 1; done

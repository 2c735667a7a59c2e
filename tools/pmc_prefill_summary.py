"""Summarises the counter passes of tools/pmc_prefill.sh for one prefill kernel into
profiles/<tag>_prefill_pmc_<kernel>.json (+ one merged CSV of per-dispatch means), with the derived
figures this repository quotes for the prefill kernels: matrix-pipe busy fraction, LDS instructions per (wave, key
tile), LDS bank conflicts, wait fractions, L2 hit rate.

    python tools/pmc_prefill_summary.py gpurun_out/pmc_pp r02
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, L, HQ = 4, 4096, 32          # workload of tools/prefill_diag.hip
FLOPS = B * 4 * HQ * 128 * (L * L - L * (L - 1) // 2)


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    means, kernel = {}, None
    for f in sorted(glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)):
        vals = {}
        for r in csv.DictReader(open(f)):
            if "prefill" not in r["Kernel_Name"]:
                continue
            kernel = r["Kernel_Name"]
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in vals.items():
            v = v[len(v) // 3:]          # drop the launches taken while the clocks ramp
            means[k] = sum(v) / len(v)
    assert kernel, "no prefill kernel in the counter files"
    short = kernel.split("lv::")[1].split("(")[0].replace("<", "_").replace(">", "").replace(", ", "_")
    m = means
    # (wave, key tile) units: a workgroup owns 256 query rows (8 waves x 32 or 4 waves x 64) and all
    # its waves walk the 64-key tiles up to the workgroup's last row together (barriers are shared)
    waves = 4 if "w64" in kernel else 8
    units = sum(waves * ((start + 256) // 64) for start in range(0, L, 256)) * B * HQ
    d = {"kernel": kernel, "workload": f"{B} x {L} tokens, {HQ}q/8kv, causal, packed (tools/prefill_p0)",
         "flops_per_launch": FLOPS, "wave_tile_units": units, "counters_mean_per_launch": m}
    g = m.get
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # SQ_VALU_MFMA_BUSY_CYCLES: shader cycles, summed over the 1024 SIMDs (= 32 x the number of
        # 32x32x16 MFMAs). SQ_BUSY_CYCLES: cycles the kernel kept a shader engine busy, summed over the
        # 32 engines (8 XCDs x 4), so /32 is the kernel's length in shader cycles at its own clock.
        kernel_cycles = m["SQ_BUSY_CYCLES"] / 32
        d["kernel_shader_cycles"] = round(kernel_cycles)
        d["mfma_pipe_busy_fraction"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * kernel_cycles), 4)
    if g("SQ_INSTS_LDS"):
        d["lds_instructions_per_wave_tile"] = round(m["SQ_INSTS_LDS"] / units, 1)
    if g("SQ_INSTS_MFMA"):
        d["mfma_instructions_per_wave_tile"] = round(m["SQ_INSTS_MFMA"] / units, 1)
    if g("SQ_INSTS_VALU"):
        d["valu_instructions_per_wave_tile"] = round(m["SQ_INSTS_VALU"] / units, 1)
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            if g(k) is not None:
                d[k.lower() + "_over_wave_cycles"] = round(m[k] / m["SQ_WAVE_CYCLES"], 4)
        d["wave_cycles_per_wave_tile"] = round(m["SQ_WAVE_CYCLES"] * 4 / units, 0)   # counter ticks every 4 cycles
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_over_lds_active"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 5)
    if g("TCC_REQ_sum"):
        d["l2_hit_rate"] = round(m["TCC_HIT_sum"] / m["TCC_REQ_sum"], 4)
    if g("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        # one count = 512 bf16 multiply-accumulate FLOP issued on the matrix pipe
        d["mfma_flops_from_mops_counter"] = m["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
        d["mfma_flops_over_algorithmic"] = round(m["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / FLOPS, 3)
    import hashlib
    d["kernel_source_sha16"] = hashlib.sha256(open(os.path.join(ROOT, "light_vllm_amd", "csrc", "prefill_attn.hip"), "rb").read()).hexdigest()[:16]
    dst = os.path.join(ROOT, "profiles", f"{tag}_prefill_pmc_{short}")
    json.dump(d, open(dst + ".json", "w"), indent=1)
    if "pp_kernel" in kernel:      # the default long-prompt kernel: what bench.py's prefill.rocprof shows
        json.dump(d, open(os.path.join(ROOT, "profiles", "prefill_counters.json"), "w"), indent=1)
    with open(dst + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Mean_Per_Launch"])
        for k in sorted(m):
            w.writerow([kernel, k, m[k]])
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()

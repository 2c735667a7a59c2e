"""Turns the two PMC passes of tools/pmc_decode.sh (FETCH_SIZE, WRITE_SIZE) into
profiles/decode_traffic.json: HBM bytes per launch of the bench's decode kernels, stamped with the
hash of the kernel source they were taken on (bench.py drops the figure when the source changes).

    python tools/pmc_decode_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE [tag]
copies the counter CSVs to profiles/<tag>_decode_pmc_{fetch,write}_size.csv as well.
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALGORITHMIC = {  # bytes per launch, bench.decode_bytes(256, 1024, Hq, 8)
    "decode_attn_kernel<2": 1_075_844_096,
    "decode_mfma_kernel<4": 1_075_844_096 + 2 * 256 * 16 * 128 * 2,
}


def load(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {dirname}"
    per_kernel = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return files[0], per_kernel


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    f_csv, fetch = load(fetch_dir, "FETCH_SIZE")
    w_csv, write = load(write_dir, "WRITE_SIZE")
    src = os.path.join(ROOT, "light_vllm_amd", "csrc", "decode_attn.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    kernels = {}
    for name, vals in fetch.items():
        f_kb = sum(vals) / len(vals)
        w_kb = sum(write.get(name, [0.0])) / max(len(write.get(name, [])), 1)
        # FETCH_SIZE counts the 128-byte requests of a 16 B/lane coalesced stream as 64 B on gfx950: x2
        hbm = int(f_kb * 1024 * 2 + w_kb * 1024)
        rec = {"fetch_size_kb_mean": f_kb, "write_size_kb_mean": w_kb, "dispatches": len(vals),
               "hbm_bytes_per_launch": hbm}
        for key, alg in ALGORITHMIC.items():
            if key in name:
                rec["algorithmic_bytes_per_launch"] = alg
                rec["ratio"] = round(hbm / alg, 4)
        kernels[name] = rec
    main_name = next((k for k in kernels if "decode_attn_kernel<2" in k), None)
    out = {"kernel": main_name, "workload": "bench.py N=1 (B=256, ctx=1024; 16q/8kv for decode_attn_kernel<2,8>, "
                                            "32q/8kv for decode_mfma_kernel<4>)",
           "kernel_source_sha16": sha,
           "correction": "FETCH_SIZE x1024 x2 (gfx950 counts 128-B requests of a 16 B/lane coalesced stream as 64 B); "
                         "WRITE_SIZE x1024 exact",
           "command": "tools/pmc_decode.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex decode_ -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux (two separate passes)",
           "kernels": kernels}
    if main_name:
        out.update(hbm_bytes_per_launch=kernels[main_name]["hbm_bytes_per_launch"],
                   algorithmic_bytes_per_launch=kernels[main_name].get("algorithmic_bytes_per_launch"))
    json.dump(out, open(os.path.join(ROOT, "profiles", "decode_traffic.json"), "w"), indent=1)
    shutil.copy(f_csv, os.path.join(ROOT, "profiles", f"{tag}_decode_pmc_fetch_size.csv"))
    shutil.copy(w_csv, os.path.join(ROOT, "profiles", f"{tag}_decode_pmc_write_size.csv"))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

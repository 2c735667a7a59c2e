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
ALGORITHMIC = [  # (kernel, label, bytes per launch = bench.decode_bytes(B, ctx, Hq, 8))
    ("decode_attn_kernel<2", "B=256 ctx=1024 16q/8kv", 1_075_844_096),
    ("decode_mfma_kernel<4", "B=256 ctx=1024 32q/8kv", 1_075_844_096 + 2 * 256 * 16 * 128 * 2),
    ("decode_mfma_kernel<4", "B=32 ctx=32768 32q/8kv (config 4 at TP=1)",
     32 * 32768 * 8 * 128 * 4 + 2 * 32 * 32 * 128 * 2 + 4 * 32 + 4 * 32 * 128),
]


def load(dirname, counter):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {dirname}"
    per_kernel = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return files[0], per_kernel       # dispatch order is the program's, the same in every pass


def classes(name, fetch_vals):
    """Launches of one kernel, split by which recorded workload they belong to (the bench runs the
    matrix-core kernel on two launches of very different size): index lists per ALGORITHMIC entry."""
    cands = [(i, a) for i, a in enumerate(ALGORITHMIC) if a[0] in name]
    out = {}
    for idx, f_kb in enumerate(fetch_vals):
        hbm = f_kb * 1024 * 2
        best = min(cands, key=lambda c: abs(hbm / c[1][2] - 1.0)) if cands else None
        if best is not None and abs(hbm / best[1][2] - 1.0) < 0.3:
            out.setdefault(best[0], []).append(idx)
    return out


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
    f_csv, fetch = load(fetch_dir, "FETCH_SIZE")
    w_csv, write = load(write_dir, "WRITE_SIZE")
    src = os.path.join(ROOT, "light_vllm_amd", "csrc", "decode_attn.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    kernels = {}
    for name, vals in fetch.items():
        wvals = write.get(name, [])
        for ci, idxs in classes(name, vals).items():
            _, label, alg = ALGORITHMIC[ci]
            f_kb = sum(vals[i] for i in idxs) / len(idxs)
            ws = [wvals[i] for i in idxs if i < len(wvals)]
            w_kb = sum(ws) / max(len(ws), 1)
            # FETCH_SIZE counts the 128-byte requests of a 16 B/lane coalesced stream as 64 B on gfx950: x2
            hbm = int(f_kb * 1024 * 2 + w_kb * 1024)
            kernels[f"{name} | {label}"] = {"fetch_size_kb_mean": f_kb, "write_size_kb_mean": w_kb, "dispatches": len(idxs),
                                           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg,
                                           "ratio": round(hbm / alg, 4)}
    main_name = next((k for k in kernels if "decode_attn_kernel<2" in k), None)
    out = {"kernel": main_name, "workload": "bench.py N=1 (B=256, ctx=1024; 16q/8kv for decode_attn_kernel<2,8>, "
                                            "32q/8kv for decode_mfma_kernel<4>)",
           "kernel_source_sha16": sha,
           "correction": "FETCH_SIZE x1024 x2 (gfx950 counts 128-B requests of a 16 B/lane coalesced stream as 64 B); "
                         "WRITE_SIZE x1024 exact",
           "command": "tools/pmc_decode.sh: rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-include-regex decode_ -- "
                      "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --aux config4 (two separate passes)",
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

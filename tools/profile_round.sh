#!/bin/bash
# Kernel-trace summaries of the bench, the prefill sweep and the engine run (rocprofv3), written
# under gpurun_out/; the *_kernel_stats.csv files are the ones copied into profiles/.
set -e
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python3 $R/bench.py --aux prefill --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_prefill -o prefill -- python3 $R/tools/kbench.py --what prefill > $R/gpurun_out/prof_prefill.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_e2e -o e2e -- python3 $R/tools/e2e_bench.py --model 8b --batch 256 --prompt-len 1024 --new-tokens 33 --modes graph > $R/gpurun_out/prof_e2e.log 2>&1
cd $R
# the bench's prefill leg times its last 50 launches (20 untimed ones come first, while the clocks ramp):
# their mean from the trace, beside the all-launch average the stats file holds
python3 - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/prof_bench/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "prefill_pp_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    timed = d[-50:]
    json.dump({"kernel": rows[0]["Kernel_Name"] if rows else None, "launches": len(d),
               "all_launches_mean_us": round(sum(d) / max(len(d), 1), 2),
               "last_50_mean_us": round(sum(timed) / max(len(timed), 1), 2),
               "last_50_min_us": round(min(timed), 2) if timed else None,
               "last_50_max_us": round(max(timed), 2) if timed else None},
              open("gpurun_out/prof_bench/prefill_timed_launches.json", "w"), indent=1)
PY
find gpurun_out/prof_bench gpurun_out/prof_prefill gpurun_out/prof_e2e -name "*kernel_stats.csv" | head
find gpurun_out -name "*kernel_trace.csv" -delete

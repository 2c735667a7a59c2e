#!/bin/bash
# One 32k-token sequence on a 4q/1kv shard (Qwen3-8B TP=8): what the split pass and the merge pass
# cost per split count, from rocprofv3 kernel traces of graph-replayed launch pairs, beside the time of
# the pair under replay. Answers whether a fused single-launch merge could beat the two launches.
# Output: gpurun_out/decode_b1_ablation.log (copy to profiles/).
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/decode_b1_ablation.log
: > $OUT
cd /tmp
for NS in 32 64 128 256 512; do
  rm -rf /tmp/b1_$NS
  KB_NS=$NS timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1_$NS -o b1 -- python3 $R/tools/kbench.py --what decode_b1 > /tmp/b1_$NS.log 2>&1 || { echo "ns=$NS failed" >> $OUT; tail -3 /tmp/b1_$NS.log >> $OUT; cat $OUT; exit 1; }
  python3 - $NS /tmp/b1_$NS /tmp/b1_$NS.log >> $OUT <<'PY'
import csv, glob, json, sys
ns, d, log = sys.argv[1], sys.argv[2], sys.argv[3]
pair = [json.loads(l) for l in open(log) if l.startswith("{")]
rows = {}
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "decode_" in n:
            rows[n.split("lv::")[-1].split("(")[0]] = (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, int(r["Calls"]))
print(json.dumps({"splits": int(ns), "pair_us_under_replay": pair[0]["us"] if pair else None,
                  "kernels_us_avg_min_calls": rows}))
PY
done
cd $R
cat $OUT

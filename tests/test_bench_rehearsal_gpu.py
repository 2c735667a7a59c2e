"""bench.py's multi-GPU control flow, rehearsed with two ranks on ONE GPU: the driver's launcher
command (python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2), rendezvous on
127.0.0.1, the direct all-reduce inside the hipGraph-captured step, max-over-ranks timing and the
JSON line. --rehearse puts both ranks on cuda:0 with a gloo group (RCCL refuses two ranks on one
device) and shrinks the workload to 4 layers x batch 8; the numbers mean nothing, the path does."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_rehearsal(gpu):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    print(json.dumps(rec)[:1500])
    assert rec["n_gpus"] == 2 and rec["rccl_ranks"] == 2 and rec["dist_backend"] == "gloo"
    assert rec["scaling"] == "strong" and rec["steps"] == 3 and rec["warmup"] == 1
    assert "config 4" in rec["config"]["workload"] and "TP=2" in rec["config"]["workload"]
    assert rec["config"]["context_len"] == 32768 and rec["config"]["rehearsal"] is True
    assert rec["config"]["hipgraph"] is True and rec["config"]["all_reduce"].startswith("direct")
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and not rec.get("comm_failed")
    assert rec["roofline"]["bytes_per_launch"] == 8 * 32768 * 4 * 128 * 4 + 2 * 8 * 16 * 128 * 2 + 4 * 8 + 4 * 8 * 128
    assert "decode_256KiB" in rec["collectives"] and "direct_us" in rec["collectives"]["decode_256KiB"]
    assert rec["config"]["all_reduce_algorithm"] == "one_shot" and rec["config"]["one_shot_inflow_bytes"] == 2 << 20
    sweep = rec["collectives"]["direct_sweep"]
    assert sweep["one_shot_inflow_bytes"] == 2 << 20 and set(sweep["sizes"]) >= {"64KiB", "2048KiB", "8192KiB"}
    assert all(v["one_shot_us"] > 0 and v["two_shot_us"] > 0 for v in sweep["sizes"].values())
    assert sweep["sizes"]["8192KiB"]["auto"] == "two_shot" and sweep["sizes"]["64KiB"]["auto"] == "one_shot"
    # one workload comparable at every N: config 4 at TP = N beside its TP = 1 base of the same run
    assert rec["scale_value"] == rec["value"] and rec["scale_ms_per_step"] == rec["ms_per_step"]
    assert "config 4" in rec["scale_workload"] and "TP=2" in rec["scale_workload"]
    assert rec["config4_tp1"]["ms_per_step"] == rec["scale_base_ms_per_step"] > 0
    assert abs(rec["scale_efficiency"] - rec["scale_base_ms_per_step"] / (2 * rec["scale_ms_per_step"])) < 1e-3

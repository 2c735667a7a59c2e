"""bench.py's accounting, checked without a GPU: the algorithmic byte counts the roofline figure
is built from, the split-count the workload runs with, and the command-line contract."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_decode_bytes_of_the_headline_workload():
    b = _bench()
    # B=256, ctx=1024, 16 q / 8 kv heads, D=128, 256-token blocks: K+V rows read once
    # (1 073 741 824 B), q in and o out (2 097 152 B), context_lens (1024 B), block-table
    # entries touched (4096 B)
    assert b.decode_bytes(256, 1024, 16, 8) == 1_075_844_096
    assert b.store_bytes(256, 8) == 2 * (256 * 8 * 128 * 2 * 2) + 4 * 256


def test_prefill_flops_of_config3():
    # 4 x 4096 tokens, 32 q heads: sum_b 4 Hq D (Lq Lk - Lq (Lq - 1) / 2) = 549.9 GFLOP
    L, Hq, D = 4096, 32, 128
    assert 4 * 4 * Hq * D * (L * L - L * (L - 1) // 2) == 549_890_031_616


def test_cli_rejects_inconsistent_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)


def test_decode_bytes_of_the_multi_gpu_workload():
    b = _bench()
    # config 4 at TP=2: 32 sequences x 32768 tokens, 16 q / 4 kv heads per rank, [32, 128] block table
    assert b.decode_bytes(32, 32768, 16, 4) == 32 * 32768 * 4 * 128 * 4 + 2 * 32 * 16 * 128 * 2 + 4 * 32 + 4 * 32 * 128
    assert b.C4["ctx"] // b.BLOCK == 128 and b.C4["hq"] == 32 and b.C4["hkv"] == 8 and b.C4["layers"] == 36


def test_recorded_traffic_is_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """The HBM-traffic figure on the bench line is a recording; it must disappear (with the reason
    on the line) as soon as decode_attn.hip is no longer the file it was recorded on."""
    import hashlib
    import json
    b = _bench()
    value, source = b.recorded_traffic("decode_attn_kernel<2", b.decode_bytes(256, 1024, 16, 8))
    src = os.path.join(ROOT, "light_vllm_amd", "csrc", "decode_attn.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    rec = json.load(open(os.path.join(ROOT, "profiles", "decode_traffic.json")))
    if rec.get("kernel_source_sha16") == sha:
        assert value == rec["hbm_bytes_per_launch"] and "recorded" in source["kind"]
    else:
        assert value is None and "stale" in source["kind"]
    assert source["current_kernel_source"] == sha


def test_recorded_traffic_matches_kernel_and_launch():
    """A recording is used only for the kernel AND launch it was taken on."""
    b = _bench()
    value, source = b.recorded_traffic("decode_attn_kernel<2", 12345)     # no such launch recorded
    assert value is None and ("no recording" in source["kind"] or "stale" in source["kind"])
    value, source = b.recorded_traffic("no_such_kernel")
    assert value is None


def test_prefill_counter_record_is_tied_to_the_kernel_source():
    import hashlib
    import json
    b = _bench()
    rec = b.recorded_prefill_counters()
    path = os.path.join(ROOT, "profiles", "prefill_counters.json")
    if not os.path.exists(path):
        assert rec is None
        return
    sha = hashlib.sha256(open(os.path.join(ROOT, "light_vllm_amd", "csrc", "prefill_attn.hip"), "rb").read()).hexdigest()[:16]
    if json.load(open(path)).get("kernel_source_sha16") == sha:
        assert 0 < rec["mfma_pipe_busy_fraction"] < 1
    else:
        assert "stale" in rec["kind"] and "mfma_pipe_busy_fraction" not in rec

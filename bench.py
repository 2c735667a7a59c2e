"""Benchmark of the attention hot path on MI355X.

Metric (BASELINE.json): paged-decode attention HBM GB/s. One "step" is one decode pass of the hot
path over every layer of the model: per layer, lv_store_kvcache of the step's new token rows, then
lv_decode_attn over the paged cache (what Attention.forward does once per layer per decode step).

Workload at N=1: BASELINE.json configs[1] — Qwen3-0.6B shapes (28 layers, 16 q / 8 kv heads, D=128),
bf16, batch 256, every sequence at 1024 tokens of context, block size 256, a seeded random physical
block permutation. At N>1 the path shards by KV head (tensor parallel rank r owns 8/N kv heads and
16/N q heads, no data-path collective inside attention); the batch grows to 256*N so per-GPU bytes
stay fixed ("weak" scaling: TP used to serve a proportionally larger batch).

`value` = algorithmic bytes of all launches of a step, summed over ranks, / max-over-ranks step time.
Inputs are resident in HBM before the timed region.

After the timed region two auxiliary measurements are attached to the same JSON line (they never
touch `value`): `collectives` times the RCCL sum all-reduce at the two message sizes the tensor-parallel
model issues (decode: [256, 4096] bf16 = 2 MiB, prefill: [16384, 4096] bf16 = 128 MiB), and `e2e`
is the engine's own decode / prefill tok/s (tools/e2e_bench.py: Qwen3-8B shapes, random weights,
TP = N, one process per GPU) run in a child process with its own time limit, so a failure there
is reported as text and cannot take the headline number with it. `--no-aux` skips both.

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
from __future__ import annotations

import argparse
import json
import os
import signal
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak
LAYERS, HQ_TOTAL, HKV_TOTAL, D, BLOCK = 28, 16, 8, 128, 256
BATCH_PER_GPU, CTX = 256, 1024


def decode_bytes(B, ctx, Hq, Hkv):
    """Algorithmic bytes of one decode launch: K and V rows of every context token once,
    q in + o out, context_lens, and the block-table entries touched."""
    nblk = (ctx + BLOCK - 1) // BLOCK
    return B * ctx * Hkv * D * 2 * 2 + 2 * B * Hq * D * 2 + 4 * B + 4 * B * nblk


def store_bytes(N, Hkv):
    return 2 * (N * Hkv * D * 2 * 2) + 4 * N


def cpu_baseline(case_cpu, Hq, Hkv, budget_s=12.0):
    """Times the CPU oracle (torch fp32 attention over block_table-gathered K/V) on one layer of
    the same workload, repeated until ~budget_s of CPU work is done."""
    from oracle.attention_oracle import decode_ref
    q, kc, vc, ctx, bt, scale = case_cpu
    B = q.size(0)
    t0 = time.perf_counter()
    decode_ref(q[:4], kc, vc, ctx[:4], bt[:4], scale)  # touch pages / warm threads
    reps, t_used = 0, 0.0
    t0 = time.perf_counter()
    while True:
        decode_ref(q, kc, vc, ctx, bt, scale)
        reps += 1
        t_used = time.perf_counter() - t0
        if t_used >= budget_s or reps >= 50:
            break
    per_call = t_used / reps
    gbs = decode_bytes(B, int(ctx[0]), Hq, Hkv) / per_call / 1e9
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle decode_ref, 1 layer of the workload (B={B}, ctx={int(ctx[0])}, Hq={Hq}, Hkv={Hkv}), "
                      f"{reps} calls, {per_call * 1e3:.1f} ms/call"}


def allreduce_bench(dist, dev, n, iters=20):
    """Average time of an in-place bf16 sum all-reduce (the collective RowParallelLinear and the
    vocab-parallel embedding issue) at decode- and prefill-sized messages; busbw = 2(n-1)/n * S / t."""
    out = {}
    for name, rows in (("decode_2MiB", 256), ("prefill_128MiB", 16384)):
        x = torch.ones(rows, 4096, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            dist.all_reduce(x)
            x.fill_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out[name] = {"avg_us": round(us, 1),
                     "busbw_GBs": round(2 * (n - 1) / n * x.numel() * 2 / (us * 1e-6) / 1e9, 1)}
        del x
    return out


TUNED_GEMMS = os.path.join(ROOT, "tools", "tunable", "tunableop.csv")   # PyTorch appends the device ordinal


MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X


def prefill_leg(torch, ops, dev, n, iters=50, warmup=20):
    """Varlen causal prefill attention of the Qwen3-8B prefill configuration (4 x 4096-token
    prompts, packed) on this rank's TP head share: TFLOP/s of the algorithmic causal FLOP count
    and the fraction of the dense bf16 matrix peak, timed with events on the launch stream.
    The untimed launches bring the clocks to what a busy engine sees: from an idle GPU the first
    ~10 ms of this matrix-bound kernel run ~19 % slower (measured: 10 launches 0.66 ms each, 100
    launches 0.555 ms each, same operands)."""
    Hq, Hkv, L, Bp = max(32 // n, 1), max(8 // n, 1), 4096, 4
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn(Bp * L, Hq, D, generator=g, device=dev).to(torch.bfloat16)
    k = torch.randn(Bp * L, Hkv, D, generator=g, device=dev).to(torch.bfloat16)
    v = torch.randn(Bp * L, Hkv, D, generator=g, device=dev).to(torch.bfloat16)
    cu = torch.arange(0, (Bp + 1) * L, L, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    for _ in range(warmup):
        ops.prefill_attention(q, k, v, cu, cu, L, L, D ** -0.5, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.prefill_attention(q, k, v, cu, cu, L, L, D ** -0.5, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = Bp * 4 * Hq * D * (L * L - L * (L - 1) // 2)
    tf = flops / (ms * 1e-3) / 1e12
    return {"workload": f"4 x 4096-token prompts, causal, packed, {Hq}q/{Hkv}kv heads, D=128",
            "launches": iters, "untimed_launches": warmup, "ms": round(ms, 4),
            "tflops": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "bound": "mfma", "flops_per_launch": flops}


def e2e_leg(n, port, mode, limit_s, tuned_gemms=False):
    """tools/e2e_bench.py in a child process group of its own (killed as a group at the limit).
    tuned_gemms: PyTorch's TunableOp picks the model's GEMM kernels from the committed selection
    file (read only, no tuning in the run; a file recorded for other library versions is ignored)."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "e2e_bench.py"), "--model", "8b", "--tp", str(n),
           "--batch", "256", "--prompt-len", "1024", "--new-tokens", "33", "--modes", mode,
           "--dist-port", str(port)]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                        "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    if tuned_gemms:
        env.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="0",
                   PYTORCH_TUNABLEOP_FILENAME=TUNED_GEMMS)
    t0 = time.perf_counter()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=limit_s)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        return {"mode": mode, "error": f"no result within {limit_s} s"}
    for line in out.splitlines():
        if line.startswith("{"):
            try:
                rec = json.loads(line)
                rec["wall_s"] = round(time.perf_counter() - t0, 1)
                rec["gemm_selection"] = "tunableop file" if tuned_gemms else "library default"
                return rec
            except ValueError:
                pass
    return {"mode": mode, "error": f"exit code {proc.returncode}: {err.strip()[-300:]}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the collectives and e2e legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    if HKV_TOTAL % n != 0:
        raise SystemExit("--gpus must divide the 8 KV heads (1, 2, 4 or 8)")

    from light_vllm_amd import ops
    import torch.distributed as dist

    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", world_size=world, rank=rank, device_id=dev)

    Hq, Hkv = HQ_TOTAL // n, HKV_TOTAL // n
    B = BATCH_PER_GPU * n
    blocks_per_seq = CTX // BLOCK
    nblk = B * blocks_per_seq
    gen = torch.Generator().manual_seed(1234 + rank)

    # one KV tensor for all layers, as ModelRunner.allocate_kv_cache lays it out
    kv = torch.empty(2, LAYERS, nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev)
    for l in range(LAYERS):
        kv[0, l].normal_()
        kv[1, l].normal_()
    q = torch.randn(B, Hq, D, device=dev).to(torch.bfloat16)
    k_new = torch.randn(B, Hkv, D, device=dev).to(torch.bfloat16)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    v_new = qkv[:, (Hq + Hkv) * D:].view(B, Hkv, D)   # V is a view into the fused activation
    perm = torch.randperm(nblk, generator=gen).to(torch.int32)
    bt_cpu = perm.view(B, blocks_per_seq).contiguous()
    ctx_cpu = torch.full((B,), CTX, dtype=torch.int32)
    slots_cpu = (bt_cpu[:, -1].long() * BLOCK + (CTX - 1) % BLOCK).to(torch.int32)
    bt, ctx, slots = bt_cpu.to(dev), ctx_cpu.to(dev), slots_cpu.to(dev)
    scale = D ** -0.5
    out = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev)
    nsplit = ops.decode_num_splits(B, Hkv, blocks_per_seq, BLOCK)
    ops.reserve_decode_workspace(dev, ops.decode_workspace_bytes(B, Hq, D, nsplit))

    def step(events=None):
        for l in range(LAYERS):
            ops.store_kvcache(k_new, v_new, kv[0, l], kv[1, l], slots)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.decode_attention(q, kv[0, l], kv[1, l], ctx, bt, scale, out=out)
            if events is not None:
                e1.record()
                events.append((e0, e1))

    def sync_all():
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = []
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    sync_all()
    elapsed = time.perf_counter() - t0
    if n > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    dec_ms = [a.elapsed_time(b) for a, b in events]
    dec_avg_s = sum(dec_ms) / len(dec_ms) / 1e3
    dbytes = decode_bytes(B, CTX, Hq, Hkv)
    step_bytes = LAYERS * (dbytes + store_bytes(B, Hkv))
    value = n * step_bytes * args.steps / elapsed / 1e9
    achieved = dbytes / dec_avg_s / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "decode_traffic.json")
    if n == 1 and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "paged-decode attn HBM GB/s",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": n,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (seeded N(0,1) bf16 q/K/V, random physical block permutation)",
        "config": {"workload": f"Qwen3-0.6B shapes TP={n}: {LAYERS} layers x (store_kvcache + paged decode), "
                               f"global batch {B}, ctx {CTX}, block {BLOCK}, per-GPU heads {Hq}q/{Hkv}kv, D={D}",
                   "global_batch": B, "context_len": CTX, "parallelism": f"tp{n}",
                   "decode_splits": nsplit},
        "pct_hbm_peak": round(100.0 * achieved / HBM_PEAK_GBS, 2),
        "roofline": {"bound": "hbm", "kernel": "decode_attn_kernel", "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "bytes_per_launch": dbytes,
                     "avg_launch_us": round(dec_avg_s * 1e6, 2)},
    }
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        case_cpu = (q.cpu(), kv[0, 0].cpu(), kv[1, 0].cpu(), ctx_cpu, bt_cpu, scale)
        result["cpu_baseline"] = cpu_baseline(case_cpu, Hq, Hkv)
    t_aux = time.perf_counter()
    if rank == 0 and not args.no_aux:
        try:
            result["prefill"] = prefill_leg(torch, ops, dev, n)
        except Exception as exc:   # reported, never fatal for the headline number
            result["prefill"] = {"error": repr(exc)[:300]}
    if n > 1 and not args.no_aux:
        try:
            result["collectives"] = allreduce_bench(dist, dev, n)
        except Exception as exc:   # reported, never fatal for the headline number
            result["collectives"] = {"error": repr(exc)[:300]}
    del kv, q, k_new, qkv, v_new, out
    torch.cuda.empty_cache()
    if n > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.no_aux:
        # every bench rank has released its GPU memory; the engine starts its own one process per GPU
        port = int(os.environ.get("MASTER_PORT", "29500")) + 17
        legs = [e2e_leg(n, port, "eager", 150)]
        if "error" not in legs[0]:
            legs.append(e2e_leg(n, port + 1, "graph", 120))
            if n == 1 and "error" not in legs[-1] and os.path.exists(TUNED_GEMMS.replace(".csv", "0.csv")):
                legs.append(e2e_leg(n, port + 2, "graph", 120, tuned_gemms=True))
        result["e2e"] = {"what": "tools/e2e_bench.py: engine generate(), Qwen3-8B shapes, random weights, "
                                 f"TP={n}, batch 256 x 1024-token prompts, 33 new tokens, greedy",
                         "runs": legs}
        result["aux_seconds"] = round(time.perf_counter() - t_aux, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

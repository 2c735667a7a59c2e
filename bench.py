"""Benchmark of the attention hot path on MI355X.

Metric (BASELINE.json): paged-decode attention HBM GB/s. One "step" is one decode pass of the hot
path over every layer of the model: per layer, lv_store_kvcache of the step's new token rows, then
lv_decode_attn over the paged cache (what Attention.forward does once per layer per decode step).

N = 1   BASELINE.json configs[1]: Qwen3-0.6B shapes (28 layers, 16 q / 8 kv heads, D=128), bf16,
        batch 256, every sequence at 1024 tokens of context, block size 256, a seeded random physical
        block permutation. `roofline` is its decode kernel (decode_attn_kernel<2,8>); `roofline_8b`
        is the same launch on Qwen3-8B TP=1 shapes (36 layers, 32 q / 8 kv heads: decode_mfma_kernel<4>),
        the configuration the 70 %-of-HBM-peak target is stated on.
N > 1   BASELINE.json configs[3]: Qwen3-8B under tensor parallelism TP = N. The path shards by KV
        head: rank r owns 8/N kv heads and 32/N q heads of every layer, a fixed GLOBAL batch of 32
        sequences at 32k tokens of context ([32, 128] block table), so the work per step is fixed
        and N divides it ("strong" scaling). What TP costs sits inside the timed step: per layer,
        after store + decode, one sum all-reduce of the [batch, 4096] bf16 activation (the o_proj
        all-reduce of RowParallelLinear, lightvllm/layers/linear.py:388-389) through the same
        layers.tp.all_reduce the model calls (direct IPC path when available, else RCCL). The 36
        layers of a step are captured in one hipGraph and the timed steps are replays.

`value` = algorithmic bytes of all attention launches of a step, summed over ranks, / max-over-ranks
step time. Inputs are resident in HBM before the timed region.

Auxiliary records on the same JSON line (never part of `value`): `cpu_baseline` (N=1: the CPU oracle
on the host cores), `prefill` (the prefill kernel on the 4 x 4096 configuration), `variants` (N=1: the
other launches the reference's call sites produce - ragged and outlier decode batches, a 32k-context
8q/1kv decode, varlen / prefix-cached / 8q/1kv prefill, the store at 16384 tokens), `config4_tp1`
(N=1: the N>1 step at TP=1, the base of the strong-scaling curve), `collectives` (N>1: all-reduce
latencies, direct path and RCCL) and `e2e` (the engine's tok/s at TP = N in child processes; a
failure there sets `e2e_failed`). `--no-aux` skips them, `--aux a,b` picks some.
At every N the line also carries `scale_value` / `scale_ms_per_step`: BASELINE config 4 at TP = N
(N = 1: the `config4_tp1` leg), one workload whose numbers are comparable across N, with
`scale_base_ms_per_step` (TP = 1, measured in the same run) and `scale_efficiency` against it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
"""
from __future__ import annotations

import argparse
import hashlib
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

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X
D, BLOCK = 128, 256
C2 = dict(name="Qwen3-0.6B", layers=28, hq=16, hkv=8, batch=256, ctx=1024)          # configs[1]
C2_8B = dict(name="Qwen3-8B", layers=36, hq=32, hkv=8, batch=256, ctx=1024)         # north-star shapes
C4 = dict(name="Qwen3-8B", layers=36, hq=32, hkv=8, batch=32, ctx=32768, hidden=4096)  # configs[3]
TUNED_GEMMS = os.path.join(ROOT, "tools", "tunable", "tunableop.csv")   # PyTorch appends the device ordinal


def decode_bytes(B, ctx, Hq, Hkv):
    """Algorithmic bytes of one decode launch: K and V rows of every context token once,
    q in + o out, context_lens, and the block-table entries touched."""
    nblk = (ctx + BLOCK - 1) // BLOCK
    return B * ctx * Hkv * D * 2 * 2 + 2 * B * Hq * D * 2 + 4 * B + 4 * B * nblk


def store_bytes(N, Hkv):
    return 2 * (N * Hkv * D * 2 * 2) + 4 * N


SETTLE_S = 1.0   # idle time between a leg's set-up and its first timed launch


class DecodeWorkload:
    """The per-layer tensors of one decode step on this rank: one KV tensor for all layers, as
    ModelRunner.allocate_kv_cache lays it out, so no layer's cache is served from the Infinity Cache
    of the layer before."""

    def __init__(self, ops, dev, layers, Hq, Hkv, B, ctx, seed):
        self.ops, self.dev, self.layers, self.Hq, self.Hkv, self.B, self.ctx = ops, dev, layers, Hq, Hkv, B, ctx
        bps = ctx // BLOCK
        nblk = B * bps
        gen = torch.Generator().manual_seed(seed)
        self.kv = torch.empty(2, layers, nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev)
        for l in range(layers):
            self.kv[0, l].normal_()
            self.kv[1, l].normal_()
        self.q = torch.randn(B, Hq, D, device=dev).to(torch.bfloat16)
        self.k_new = torch.randn(B, Hkv, D, device=dev).to(torch.bfloat16)
        qkv = torch.randn(B, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
        self.v_new = qkv[:, (Hq + Hkv) * D:].view(B, Hkv, D)   # V is a view into the fused activation
        self.bt_cpu = torch.randperm(nblk, generator=gen).to(torch.int32).view(B, bps).contiguous()
        self.ctx_cpu = torch.full((B,), ctx, dtype=torch.int32)
        slots = (self.bt_cpu[:, -1].long() * BLOCK + (ctx - 1) % BLOCK).to(torch.int32)
        self.bt, self.ctx_t, self.slots = self.bt_cpu.to(dev), self.ctx_cpu.to(dev), slots.to(dev)
        self.scale = D ** -0.5
        self.out = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=dev)
        self.nsplit = ops.decode_num_splits(B, Hkv, bps, BLOCK)
        ops.reserve_decode_workspace(dev, ops.decode_workspace_bytes(B, Hq, D, self.nsplit))
        self.decode_bytes = decode_bytes(B, ctx, Hq, Hkv)
        self.step_bytes = layers * (self.decode_bytes + store_bytes(B, Hkv))
        # For up to ~0.8 s after a large hipFree or fill (the previous leg's caches, this leg's random
        # numbers) the same launch on the same buffers runs 3-5 % slower, then returns to the time it
        # holds for thousands of back-to-back launches (tools/decode_state_probe.py). Every leg
        # starts from that settled state, not from the tail of its own set-up.
        torch.cuda.synchronize()
        time.sleep(SETTLE_S)

    def step(self, events=None, after_layer=None):
        ops, kv = self.ops, self.kv
        for l in range(self.layers):
            ops.store_kvcache(self.k_new, self.v_new, kv[0, l], kv[1, l], self.slots)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            ops.decode_attention(self.q, kv[0, l], kv[1, l], self.ctx_t, self.bt, self.scale, out=self.out)
            if events is not None:
                e1.record()
                events.append((e0, e1))
            if after_layer is not None:
                after_layer()

    def roofline(self, steps, kernel):
        """Events around every decode launch of `steps` eager passes, on the launch stream."""
        events = []
        self.step()
        torch.cuda.synchronize()
        for _ in range(steps):
            self.step(events)
        torch.cuda.synchronize()
        times = sorted(a.elapsed_time(b) for a, b in events)
        avg_s = sum(times) / len(times) / 1e3
        achieved = self.decode_bytes / avg_s / 1e9
        return {"bound": "hbm", "kernel": kernel, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "bytes_per_launch": self.decode_bytes, "avg_launch_us": round(avg_s * 1e6, 2),
                "median_launch_us": round(times[len(times) // 2] * 1e3, 2), "max_launch_us": round(times[-1] * 1e3, 2),
                "launches_timed": len(events),
                "shape": f"B={self.B} ctx={self.ctx} {self.Hq}q/{self.Hkv}kv x {self.layers} layers, splits {self.nsplit}"}

    def free(self):
        del self.kv, self.q, self.k_new, self.v_new, self.out
        torch.cuda.empty_cache()


def _sha16(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]


def recorded_traffic(kernel="decode_attn_kernel<2", launch_bytes=None):
    """HBM bytes per launch of a decode kernel from the PMC passes of tools/pmc_decode.sh
    (profiles/decode_traffic.json). A recording, not a measurement of this run: returned only while
    the kernel source it was taken on is the one in the tree. launch_bytes: the algorithmic bytes of
    the launch in question; the recording of the same kernel whose algorithmic bytes match is used."""
    path = os.path.join(ROOT, "profiles", "decode_traffic.json")
    src = os.path.join(ROOT, "light_vllm_amd", "csrc", "decode_attn.hip")
    try:
        rec = json.load(open(path))
        sha = _sha16(src)
    except (OSError, ValueError):
        return None, {"kind": "none"}
    source = {"kind": "recorded PMC passes (FETCH_SIZE, WRITE_SIZE), not collected in this run",
              "file": "profiles/decode_traffic.json",
              "recorded_on_kernel_source": rec.get("kernel_source_sha16"), "current_kernel_source": sha}
    if rec.get("kernel_source_sha16") != sha:
        source["kind"] = "stale: recorded on another version of decode_attn.hip, dropped"
        return None, source
    for name, k in rec.get("kernels", {}).items():
        if kernel in name and (launch_bytes is None or k.get("algorithmic_bytes_per_launch") == launch_bytes):
            source["kernel"] = name
            return k.get("hbm_bytes_per_launch"), source
    source["kind"] = "no recording of this kernel on this launch"
    return None, source


def recorded_prefill_counters():
    """Matrix-pipe utilisation of the prefill kernel from rocprofv3 counters (tools/pmc_prefill.sh +
    tools/pmc_prefill_summary.py on the 4 x 4096, 32q/8kv launch): a recording, shown only while
    prefill_attn.hip is the source it was taken on."""
    path = os.path.join(ROOT, "profiles", "prefill_counters.json")
    src = os.path.join(ROOT, "light_vllm_amd", "csrc", "prefill_attn.hip")
    try:
        rec = json.load(open(path))
        sha = _sha16(src)
    except (OSError, ValueError):
        return None
    out = {"kind": "recorded rocprofv3 --pmc passes, not collected in this run", "file": "profiles/prefill_counters.json",
           "recorded_on_kernel_source": rec.get("kernel_source_sha16"), "current_kernel_source": sha}
    if rec.get("kernel_source_sha16") != sha:
        out["kind"] = "stale: recorded on another version of prefill_attn.hip, dropped"
        return out
    for key in ("kernel", "mfma_pipe_busy_fraction", "mfma_flops_over_algorithmic", "l2_hit_rate",
                "lds_instructions_per_wave_tile", "mfma_instructions_per_wave_tile", "kernel_shader_cycles"):
        if key in rec:
            out[key] = rec[key]
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(w, budget_s=10.0):
    """The CPU oracle (torch fp32 attention over block_table-gathered K/V, the reference's CPU
    plumbing path of BASELINE config 1) on one layer of the config-2 workload, on the threads this
    job owns; then a single-thread leg on 32 of the 256 sequences, and the config-3 prefill on
    1 x 4096 tokens (a quarter of its batch), each bounded in time."""
    from oracle.attention_oracle import decode_ref, prefill_ref
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    q, kc, vc = w.q.cpu(), w.kv[0, 0].cpu(), w.kv[1, 0].cpu()
    ctx, bt, scale = w.ctx_cpu, w.bt_cpu, w.scale

    def timed(fn, budget, max_reps):
        fn()
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            used = time.perf_counter() - t0
            if used >= budget or reps >= max_reps:
                return used / reps, reps

    per_call, reps = timed(lambda: decode_ref(q, kc, vc, ctx, bt, scale), budget_s, 50)
    rec = {"value": round(w.decode_bytes / per_call / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
           "sample": f"oracle decode_ref, 1 layer of the workload (B={w.B}, ctx={w.ctx}, Hq={w.Hq}, Hkv={w.Hkv}), "
                     f"{reps} calls, {per_call * 1e3:.1f} ms/call",
           "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(),
           "threads_note": "threads = this job's CPU share (OMP_NUM_THREADS), not the host's core count"}
    torch.set_num_threads(1)
    nb = 32
    per1, reps1 = timed(lambda: decode_ref(q[:nb], kc, vc, ctx[:nb], bt[:nb], scale), 4.0, 10)
    rec["single_thread"] = {"value": round(decode_bytes(nb, w.ctx, w.Hq, w.Hkv) / per1 / 1e9, 3), "unit": "GB/s",
                            "cores": 1, "sample": f"first {nb} sequences of the same layer, {reps1} calls, {per1 * 1e3:.1f} ms/call"}
    torch.set_num_threads(threads)
    L, Hq, Hkv = 4096, 32, 8
    g = torch.Generator().manual_seed(3)
    pq = torch.randn(L, Hq, D, generator=g).to(torch.bfloat16)
    pk = torch.randn(L, Hkv, D, generator=g).to(torch.bfloat16)
    pv = torch.randn(L, Hkv, D, generator=g).to(torch.bfloat16)
    cu = torch.tensor([0, L], dtype=torch.int32)
    perp, repsp = timed(lambda: prefill_ref(pq, pk, pv, cu, cu, scale), 4.0, 10)
    flops = 4 * Hq * D * (L * L - L * (L - 1) // 2)
    rec["prefill"] = {"value": round(flops / perp / 1e12, 4), "unit": "TFLOP/s", "cores": threads,
                      "sample": f"oracle prefill_ref, 1 x {L} tokens of the 4 x 4096 configuration ({Hq}q/{Hkv}kv), "
                                f"{repsp} calls, {perp * 1e3:.0f} ms/call"}
    return rec


def prefill_leg(ops, dev, n, iters=50, warmup=20):
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


def _event_time(fn, iters, warmup=2):
    """Mean seconds per call of fn, events on the launch stream."""
    for _ in range(warmup):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def ragged_decode_leg(ops, dev, Hq, Hkv, lens, how, layers=6, reps=3):
    """One paged-decode launch per layer cache on a batch of unequal lengths (python ints), blocks
    handed out as the block manager would (ceil(len / 256) per sequence, random physical order).
    how: "ordered" = lv_decode_attn with the longest-first launch order ModelRunner.prepare_decode
    passes, "planned" = lv_decode_attn_planned on the work list decode_plan.build_plan cuts."""
    from light_vllm_amd.decode_plan import build_plan
    B = len(lens)
    need = [-(-n // BLOCK) for n in lens]
    nblk, width = sum(need), max(need)
    perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(7)).to(torch.int32)
    bt = torch.zeros(B, width, dtype=torch.int32)
    pos = 0
    for i, n in enumerate(need):
        bt[i, :n] = perm[pos:pos + n]
        pos += n
    kv = torch.empty(layers, 2, nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev).normal_()
    q = torch.randn(B, Hq, D, device=dev).to(torch.bfloat16)
    out = torch.empty_like(q)
    ctx_t, bt_d = torch.tensor(lens, dtype=torch.int32, device=dev), bt.to(dev)
    order = plan = None
    if how == "ordered":
        order = torch.tensor(sorted(range(B), key=lambda j: -lens[j]), dtype=torch.int32, device=dev)
    else:
        plan = build_plan(lens, Hkv, Hq // Hkv).to(dev)
    state = {"i": 0}

    def fn():
        l = state["i"] % layers
        state["i"] += 1
        ops.decode_attention(q, kv[l, 0], kv[l, 1], ctx_t, bt_d, D ** -0.5, num_splits=1 if how == "ordered" else 0,
                             out=out, seq_order=order, plan=plan)
    torch.cuda.synchronize()
    time.sleep(SETTLE_S)
    t = _event_time(fn, layers * reps, warmup=layers)
    nbytes = sum(lens) * Hkv * D * 4 + 2 * B * Hq * D * 2 + 4 * B + 4 * nblk
    rec = {"how": how, "heads": f"{Hq}q/{Hkv}kv", "batch": B, "tokens": sum(lens), "longest": max(lens),
           "bytes_per_launch": nbytes, "avg_launch_us": round(t * 1e6, 2), "achieved": round(nbytes / t / 1e9, 1),
           "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm"}
    if plan is not None:
        rec["pieces"], rec["piece_tokens"], rec["most_pieces_per_sequence"] = plan.used, plan.piece_tokens, plan.max_splits
    del kv
    torch.cuda.empty_cache()
    return rec


def prefill_variant_leg(ops, dev, lens, Hq, Hkv, cached=0, iters=30, warmup=20):
    """Prefill attention on prompts of `lens` tokens whose first `cached` tokens each are already in
    the paged cache (0: packed K/V, no block table), with the query-tile launch order the runner
    builds from the lengths (ops.prefill_tile_order; None for 8 kv heads)."""
    lens_q = [n - cached for n in lens]
    tq = sum(lens_q)
    q = torch.randn(tq, Hq, D, device=dev).to(torch.bfloat16)
    out = torch.empty_like(q)
    cu_q = torch.tensor([0] + torch.tensor(lens_q).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    cu_k = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    order = ops.prefill_tile_order(lens_q, lens, Hq, Hkv, dev, paged=cached > 0)
    if cached:
        need = [-(-n // BLOCK) for n in lens]
        nblk = sum(need)
        kc = torch.randn(nblk, BLOCK, Hkv, D, device=dev).to(torch.bfloat16)
        vc = torch.randn(nblk, BLOCK, Hkv, D, device=dev).to(torch.bfloat16)
        perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(11)).to(torch.int32)
        bt = torch.full((len(lens), max(need)), -1, dtype=torch.int32)
        pos = 0
        for i, n in enumerate(need):
            bt[i, :n] = perm[pos:pos + n]
            pos += n
        bt = bt.to(dev)
        fn = lambda: ops.prefill_attention(q, kc, vc, cu_q, cu_k, max(lens_q), max(lens), D ** -0.5,   # noqa: E731
                                           block_table=bt, out=out, tile_order=order)
    else:
        k = torch.randn(tq, Hkv, D, device=dev).to(torch.bfloat16)
        v = torch.randn(tq, Hkv, D, device=dev).to(torch.bfloat16)
        fn = lambda: ops.prefill_attention(q, k, v, cu_q, cu_k, max(lens_q), max(lens), D ** -0.5,     # noqa: E731
                                           out=out, tile_order=order)
    t = _event_time(fn, iters, warmup)
    flops = sum(4 * Hq * D * (lq * lk - lq * (lq - 1) // 2) for lq, lk in zip(lens_q, lens))
    return {"prompts": lens, "cached_tokens_each": cached, "heads": f"{Hq}q/{Hkv}kv", "ms": round(t * 1e3, 4),
            "tflops": round(flops / t / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "frac": round(flops / t / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "bound": "mfma", "flops_per_launch": flops, "tile_order": order is not None}


def store_leg(ops, dev, N=16384, Hq=32, Hkv=8, caches=4):
    """lv_store_kvcache at a prefill's size: N rows into random distinct slots, V a strided view into
    the fused qkv activation as the model hands it over."""
    qkv = torch.randn(N, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    k = qkv[:, Hq * D:(Hq + Hkv) * D].contiguous().view(N, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D:].view(N, Hkv, D)
    nblk = 2 * N // BLOCK
    kv = torch.zeros(caches, 2, nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev)
    slots = torch.randperm(nblk * BLOCK, device=dev)[:N].to(torch.int32)
    state = {"i": 0}

    def fn():
        c = state["i"] % caches
        state["i"] += 1
        ops.store_kvcache(k, v, kv[c, 0], kv[c, 1], slots)
    t = _event_time(fn, 40, warmup=8)
    nbytes = store_bytes(N, Hkv)
    return {"tokens": N, "heads_kv": Hkv, "avg_launch_us": round(t * 1e6, 2), "bytes_per_launch": nbytes,
            "achieved": round(nbytes / t / 1e9, 1), "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4), "bound": "hbm"}


def variants_leg(ops, dev, sync_all):
    """The other launches the reference's two attention call sites and its store produce
    (lightvllm/layers/attention.py:104-108, 159-165, 172-175; engine/model_runner.py:369-370 for the
    prefix-cached mode), each with its own roofline fraction. Every entry stands alone: a failure
    is recorded in place."""
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:   # noqa: BLE001 - reported, never fatal for the headline number
            out[name] = {"error": repr(exc)[:300]}
        torch.cuda.empty_cache()

    ragged = torch.randint(1, 1025, (256,), generator=torch.Generator().manual_seed(0)).tolist()
    leg("decode_ragged_16q8kv", lambda: ragged_decode_leg(ops, dev, 16, 8, ragged, "ordered"))
    leg("decode_ragged_32q8kv", lambda: ragged_decode_leg(ops, dev, 32, 8, ragged, "ordered"))
    outl = [16384] * 4 + ragged[4:]
    leg("decode_outliers_32q8kv_one_split_count", lambda: ragged_decode_leg(ops, dev, 32, 8, outl, "ordered"))
    leg("decode_outliers_32q8kv_work_list", lambda: ragged_decode_leg(ops, dev, 32, 8, outl, "planned"))

    def c5_decode():
        w = DecodeWorkload(ops, dev, 4, 8, 1, 64, 32768, seed=77)
        w.step()
        r = w.roofline(5, "decode_mfma_kernel<8>")
        w.free()
        return r
    leg("decode_c5_8q1kv_b64_ctx32k", c5_decode)
    leg("prefill_varlen", lambda: prefill_variant_leg(ops, dev, [4096, 3000, 1024, 8264], 32, 8))
    leg("prefill_prefix_1024_cached", lambda: prefill_variant_leg(ops, dev, [4096] * 4, 32, 8, cached=1024))
    leg("prefill_c5_8q1kv", lambda: prefill_variant_leg(ops, dev, [4096] * 4, 8, 1))
    leg("prefill_c5_8q1kv_prefix_1024_cached", lambda: prefill_variant_leg(ops, dev, [4096] * 4, 8, 1, cached=1024))
    leg("store_16384_tokens", lambda: store_leg(ops, dev))
    return out


def sustained_mfma_peak():
    """tools/mfma_peak (built by __graft_entry__.build): every SIMD issues bf16 MFMAs back to back
    from registers for ~50 ms, operands random like attention's. What it reaches is the matrix rate
    the chip sustains at the clock its power management allows under that load, the practical
    ceiling under the 2.5 PFLOP/s the spec quotes at 2.4 GHz. None when the tool is not built."""
    exe = os.path.join(ROOT, "tools", "mfma_peak")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, "2", "1"], capture_output=True, text=True, timeout=60).stdout
        for line in out.splitlines():
            if line.startswith("RESULT "):
                return json.loads(line[len("RESULT "):])
    except (OSError, ValueError, subprocess.TimeoutExpired):
        pass
    return None


def time_all_reduce(fn, x, iters=50):
    for _ in range(5):
        fn(x)
        x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn(x)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def collectives_leg(dist, dev, n, direct):
    """Average time of the in-place bf16 sum all-reduce of the TP model at its decode-sized
    messages ([32, 4096] and [256, 4096]) on the direct IPC path and on RCCL, and at the
    prefill-sized message ([16384, 4096], RCCL only); busbw = 2(n-1)/n * bytes / t. Then a sweep of
    the direct path from 64 KiB to 8 MiB under both of its algorithms, from which the one-shot /
    two-shot threshold (lv_comm_set_one_shot_inflow) is read off after a run across GPUs.
    A failure of the direct path on this rank is recorded per message and never skips a later
    RCCL call: every rank issues the same torch.distributed collectives in the same order."""
    out = {}

    def direct_time(x, algorithm):
        us = time_all_reduce(lambda t: direct.all_reduce(t, algorithm), x)
        if direct.failed():
            raise RuntimeError("a peer did not arrive within the communicator's timeout")
        return us

    for name, rows in (("decode_256KiB", 32), ("decode_2MiB", 256), ("prefill_128MiB", 16384)):
        x = torch.ones(rows, 4096, dtype=torch.bfloat16, device=dev)
        rec = {}
        if dist.get_backend() == "nccl":
            us = time_all_reduce(dist.all_reduce, x, iters=50 if rows <= 256 else 10)
            rec["rccl_us"] = round(us, 1)
            rec["rccl_busbw_GBs"] = round(2 * (n - 1) / n * x.numel() * 2 / (us * 1e-6) / 1e9, 1)
        if direct is not None and direct.usable(x):
            try:
                us = direct_time(x, "auto")
                rec["direct_us"] = round(us, 1)
                rec["direct_algorithm"] = direct.algorithm_for(x.numel() * 2)
                rec["direct_busbw_GBs"] = round(2 * (n - 1) / n * x.numel() * 2 / (us * 1e-6) / 1e9, 1)
            except Exception as exc:   # noqa: BLE001 - recorded; the RCCL sequence below stays in step
                rec["direct_error"] = repr(exc)[:200]
                torch.cuda.synchronize()
        out[name] = rec
        del x
    if direct is not None:
        sweep = {"one_shot_inflow_bytes": direct.one_shot_inflow, "sizes": {}}
        for kib in (64, 128, 256, 512, 1024, 2048, 4096, 8192):
            x = torch.ones(kib * 512, dtype=torch.bfloat16, device=dev)
            rec = {"auto": direct.algorithm_for(kib * 1024)}
            try:
                for algorithm in ("one_shot", "two_shot"):
                    rec[algorithm + "_us"] = round(direct_time(x, algorithm), 1)
            except Exception as exc:   # noqa: BLE001
                rec["error"] = repr(exc)[:200]
                torch.cuda.synchronize()
                sweep["sizes"][f"{kib}KiB"] = rec
                break
            sweep["sizes"][f"{kib}KiB"] = rec
            del x
        out["direct_sweep"] = sweep
    return out


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


def tp_step_setup(ops, dev, n, rank, cfg, layers, batch):
    """The config-4 step on this rank: workload + the activation the per-layer all-reduce sums."""
    w = DecodeWorkload(ops, dev, layers, cfg["hq"] // n, cfg["hkv"] // n, batch, cfg["ctx"], seed=4321 + rank)
    hidden = torch.ones(batch, cfg["hidden"], dtype=torch.bfloat16, device=dev)
    return w, hidden


def run_steps(step, steps, warmup, sync_all, use_graph):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides. With use_graph the
    step is captured once and the timed steps are replays. Returns (elapsed seconds, graph used, note)."""
    note = None
    graph = None
    if use_graph:
        try:
            step()                                   # warm run outside capture
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
        except Exception as exc:                      # reported on the line; the eager step still measures
            note = f"hipGraph capture failed, eager steps timed instead: {exc!r}"[:300]
            graph = None
            torch.cuda.synchronize()
    run = graph.replay if graph is not None else step
    for _ in range(warmup):
        run()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    sync_all()
    return time.perf_counter() - t0, graph is not None, note


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-aux", action="store_true", help="skip the auxiliary legs")
    ap.add_argument("--aux", default="prefill,variants,config4,collectives,e2e",
                    help="comma list of auxiliary legs to run (prefill, variants, config4, collectives, e2e)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on ONE GPU: every rank on cuda:0, gloo group, 4 layers, batch 8 "
                         "(proves launcher, rendezvous, graph and JSON path; not a measurement)")
    args = ap.parse_args()

    aux = set() if args.no_aux else {a for a in args.aux.split(",") if a}
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        raise SystemExit(f"--gpus {n} but WORLD_SIZE={world}")
    if C4["hkv"] % n != 0:
        raise SystemExit("--gpus must divide the 8 KV heads (1, 2, 4 or 8)")

    from light_vllm_amd import ops
    import torch.distributed as dist

    dev = torch.device("cuda:0" if args.rehearse else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)
    direct = None
    backend = None
    if n > 1:
        from light_vllm_amd import comm as lvcomm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", world_size=world, rank=rank)
        else:
            dist.init_process_group("nccl", world_size=world, rank=rank, device_id=dev)
        backend = dist.get_backend()
        direct = lvcomm.create_for_group(dev)
        lvcomm.install(direct)

    def sync_all():
        torch.cuda.synchronize()
        if n > 1:
            dist.barrier()
            torch.cuda.synchronize()

    result = {"metric": "paged-decode attn HBM GB/s", "unit": "GB/s", "n_gpus": n, "steps": args.steps,
              "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None, "dtype": "bf16",
              "data": "synthetic (seeded N(0,1) bf16 q/K/V, random physical block permutation)",
              "settle_s": SETTLE_S}

    if n == 1:
        c = C2
        w = DecodeWorkload(ops, dev, c["layers"], c["hq"], c["hkv"], c["batch"], c["ctx"], seed=1234)
        elapsed, _, _ = run_steps(w.step, args.steps, args.warmup, sync_all, use_graph=False)
        value = w.step_bytes * args.steps / elapsed / 1e9
        roof = w.roofline(args.steps, "decode_attn_kernel<2,8>")
        roof["traffic"], roof["traffic_source"] = recorded_traffic("decode_attn_kernel<2", w.decode_bytes)
        result.update({
            "value": round(value, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "scaling": "weak",
            "config": {"workload": f"BASELINE config 2: {c['name']} shapes TP=1: {c['layers']} layers x (store_kvcache + "
                                   f"paged decode), global batch {c['batch']}, ctx {c['ctx']}, block {BLOCK}, "
                                   f"heads {c['hq']}q/{c['hkv']}kv, D={D}, eager launches",
                       "global_batch": c["batch"], "context_len": c["ctx"], "parallelism": "tp1",
                       "decode_splits": w.nsplit},
            "pct_hbm_peak": round(100.0 * roof["frac"], 2), "roofline": roof})
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w)
        w.free()
        # the same launch on Qwen3-8B TP=1 shapes: the matrix-core decode kernel
        c8 = C2_8B
        w8 = DecodeWorkload(ops, dev, c8["layers"], c8["hq"], c8["hkv"], c8["batch"], c8["ctx"], seed=1235)
        el8, _, _ = run_steps(w8.step, max(args.steps // 2, 2), 2, sync_all, use_graph=False)
        r8 = w8.roofline(max(args.steps // 2, 2), "decode_mfma_kernel<4>")
        r8["step_GBs"] = round(w8.step_bytes * max(args.steps // 2, 2) / el8 / 1e9, 2)
        r8["workload"] = (f"{c8['name']} shapes TP=1: {c8['layers']} layers x (store_kvcache + paged decode), batch "
                          f"{c8['batch']}, ctx {c8['ctx']}, heads {c8['hq']}q/{c8['hkv']}kv")
        r8["traffic"], r8["traffic_source"] = recorded_traffic("decode_mfma_kernel<4", w8.decode_bytes)
        result["roofline_8b"] = r8
        w8.free()
    else:
        c = C4
        layers, batch = (4, 8) if args.rehearse else (c["layers"], c["batch"])
        from light_vllm_amd.layers import tp
        w, hidden = tp_step_setup(ops, dev, n, rank, c, layers, batch)
        use_graph = direct is not None or backend == "nccl"
        cpu_or_dev = dev if backend == "nccl" else "cpu"

        def timed_step():
            """(max elapsed over ranks, graph used, note, failed on any rank). A rank whose direct
            all-reduce gives up (bounded waits: every rank's kernels end) reports it instead of
            leaving; the ranks then agree through torch.distributed."""
            failed, el, graphed, note = 0.0, 0.0, False, None
            try:
                el, graphed, note = run_steps(lambda: w.step(after_layer=lambda: tp.all_reduce(hidden)),
                                              args.steps, args.warmup, sync_all, use_graph)
                if direct is not None and lvcomm.active() is not None and direct.failed():
                    raise RuntimeError("direct all-reduce: a peer did not arrive within the timeout")
            except Exception as exc:   # noqa: BLE001
                failed, note = 1.0, f"rank {rank}: {exc!r}"[:300]
                torch.cuda.synchronize()
            t = torch.tensor([el, failed], dtype=torch.float64, device=cpu_or_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0].item()), graphed, note, bool(t[1].item() > 0)

        elapsed, graphed, note, bad = timed_step()
        direct_fell_back = False
        if bad and direct is not None and backend == "nccl":
            # the direct transport failed on some rank: time the same step on RCCL instead
            lvcomm.install(None)
            direct_fell_back = True
            hidden.fill_(1.0)
            elapsed, graphed, note2, bad = timed_step()
            note = f"direct all-reduce failed ({note}); re-timed on RCCL" + (f"; {note2}" if note2 else "")
        if bad:
            raise SystemExit(f"bench: the timed step failed on some rank: {note}")
        value = n * w.step_bytes * args.steps / elapsed / 1e9
        kern = "decode_mfma_kernel<4>"
        roof = w.roofline(2, kern)
        msg = batch * c["hidden"] * 2
        result.update({
            "value": round(value, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 4), "scaling": "strong",
            "config": {"workload": f"BASELINE config 4: {c['name']} shapes TP={n}: {layers} layers x (store_kvcache + "
                                   f"paged decode + all-reduce of [{batch}, {c['hidden']}] bf16), global batch {batch}, "
                                   f"ctx {c['ctx']}, block table [{batch}, {c['ctx'] // BLOCK}], per-rank heads "
                                   f"{w.Hq}q/{w.Hkv}kv, D={D}, " + ("hipGraph replay" if graphed else "eager launches"),
                       "global_batch": batch, "context_len": c["ctx"], "parallelism": f"tp{n}",
                       "decode_splits": w.nsplit, "hipgraph": graphed, "rehearsal": bool(args.rehearse),
                       "all_reduce": "direct (lv_comm over IPC-mapped peer buffers)"
                                     if direct is not None and not direct_fell_back else backend,
                       "all_reduce_bytes": msg,
                       "all_reduce_algorithm": direct.algorithm_for(msg)
                                               if direct is not None and not direct_fell_back else None,
                       "one_shot_inflow_bytes": direct.one_shot_inflow if direct is not None else None},
            "rccl_ranks": dist.get_world_size(), "dist_backend": backend,
            "pct_hbm_peak": round(100.0 * roof["frac"], 2), "roofline": roof})
        if note:
            result["config"]["graph_note"] = note
        if direct is not None and (direct_fell_back or direct.failed()):
            result["comm_failed"] = True
        del hidden
        w.free()

    t_aux = time.perf_counter()
    if rank == 0 and "prefill" in aux:
        try:
            result["prefill"] = prefill_leg(ops, dev, n)
            burn = sustained_mfma_peak()
            if burn:
                result["prefill"]["sustained_mfma"] = burn
                result["prefill"]["frac_of_sustained"] = round(result["prefill"]["tflops"] / burn["tflops"], 4)
            if n == 1:
                rp = recorded_prefill_counters()
                if rp:
                    result["prefill"]["rocprof"] = rp
        except Exception as exc:   # reported, never fatal for the headline number
            result["prefill"] = {"error": repr(exc)[:300]}
    if n == 1 and "variants" in aux:
        result["variants"] = variants_leg(ops, dev, sync_all)
    scale_base = None
    if "config4" in aux and (n == 1 or rank == 0):
        # the multi-GPU step at TP=1 (no all-reduce): the base of the strong-scaling curve
        # (a rehearsal keeps its reduced size: 4 layers, batch 8)
        try:
            l4, b4 = (4, 8) if args.rehearse else (C4["layers"], C4["batch"])
            w4, _ = tp_step_setup(ops, dev, 1, 0, C4, l4, b4)
            el4, graphed, note = run_steps(w4.step, 5, 2, lambda: torch.cuda.synchronize(), use_graph=True)
            r4 = w4.roofline(1, "decode_mfma_kernel<4>")
            r4["traffic"], r4["traffic_source"] = recorded_traffic("decode_mfma_kernel<4", w4.decode_bytes)
            result["config4_tp1"] = {"workload": f"BASELINE config 4 at TP=1: {l4} layers x (store_kvcache + paged "
                                                 f"decode), batch {b4}, ctx {C4['ctx']}, heads {C4['hq']}q/{C4['hkv']}kv",
                                     "value": round(w4.step_bytes * 5 / el4 / 1e9, 2), "unit": "GB/s",
                                     "ms_per_step": round(el4 / 5 * 1e3, 4), "hipgraph": graphed, "roofline": r4}
            scale_base = result["config4_tp1"]
            w4.free()
        except Exception as exc:
            result["config4_tp1"] = {"error": repr(exc)[:300]}
    if n > 1:
        sync_all()      # the other ranks wait here (process-group barrier) while rank 0 measures the TP = 1 base
    # One workload whose numbers are comparable at every N: BASELINE config 4 (global batch 32 at 32k
    # context) at TP = N. N = 1: the leg above; N > 1: the headline itself, beside the TP = 1 step
    # measured by rank 0 of this same run.
    if n == 1 and scale_base is not None:
        result.update(scale_value=scale_base["value"], scale_ms_per_step=scale_base["ms_per_step"])
    elif n > 1:
        result.update(scale_value=result["value"], scale_ms_per_step=result["ms_per_step"])
    if "scale_value" in result:
        result["scale_workload"] = (f"BASELINE config 4: Qwen3-8B shapes, global batch {8 if args.rehearse else C4['batch']}, "
                                    f"ctx {C4['ctx']}, TP={n}, hipGraph replay" + (" (rehearsal size)" if args.rehearse else ""))
        if scale_base is not None:
            result["scale_base_ms_per_step"] = scale_base["ms_per_step"]
            result["scale_efficiency"] = round(scale_base["ms_per_step"] / (n * result["scale_ms_per_step"]), 4)
    if n > 1 and "collectives" in aux:
        try:
            result["collectives"] = collectives_leg(dist, dev, n, None if direct_fell_back else direct)
        except Exception as exc:
            result["collectives"] = {"error": repr(exc)[:300]}
    torch.cuda.empty_cache()
    if n > 1:
        from light_vllm_amd import comm as lvcomm
        dist.barrier()
        lvcomm.install(None)
        if direct is not None:
            direct.close()
        dist.destroy_process_group()
    if rank == 0 and "e2e" in aux and not args.rehearse:
        # every bench rank has released its GPU memory; the engine starts its own one process per GPU
        port = int(os.environ.get("MASTER_PORT", "29500")) + 17
        legs = [e2e_leg(n, port, "eager", 150)]
        if "error" not in legs[0]:
            legs.append(e2e_leg(n, port + 1, "graph", 150))
            if n == 1 and "error" not in legs[-1] and os.path.exists(TUNED_GEMMS.replace(".csv", "0.csv")):
                legs.append(e2e_leg(n, port + 2, "graph", 120, tuned_gemms=True))
        result["e2e"] = {"what": "tools/e2e_bench.py: engine generate(), Qwen3-8B shapes, random weights, "
                                 f"TP={n}, batch 256 x 1024-token prompts, 33 new tokens, greedy",
                         "runs": legs}
        result["e2e_failed"] = any("error" in leg for leg in legs)
        result["aux_seconds"] = round(time.perf_counter() - t_aux, 1)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

"""Kernel-level sweep on one MI355X: achieved algorithmic GB/s (decode, store) or TFLOP/s (prefill)
per shape, timed with events on torch's current stream. Not the contract bench (see bench.py)."""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from light_vllm_amd import ops  # noqa: E402

D, BLOCK = 128, 256
dev = torch.device("cuda:0")


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def timeit_graph(fn, iters=20, warmup=3, replays=5):
    """The same launches captured in one hipGraph and replayed: takes the host's per-call cost
    (argument checks, ctypes) out of cases whose GPU time is shorter than it."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * replays) * 1e-3


def decode_case(B, ctx, Hq, Hkv, nsplit=0, layers=4, ragged=False, ordered=False, graph=False, data="normal",
                plan=None, outliers=()):
    """outliers: lengths that replace the first sequences' (a few long contexts in a batch of short ones).
    plan: None, or keyword arguments of decode_plan.build_plan (a host-built work list: long
    sequences cut into more pieces than short ones)."""
    bps = (max([ctx, *outliers]) + BLOCK - 1) // BLOCK
    nblk = B * bps if not outliers else B * ((ctx + BLOCK - 1) // BLOCK) + len(outliers) * bps
    # several distinct caches so the working set exceeds the 256 MiB Infinity Cache
    per_layer = 2 * nblk * BLOCK * Hkv * D * 2
    layers = max(layers, min(64, (1 << 30) // max(per_layer, 1) + 1))
    kv = torch.empty(layers, 2, nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev)
    if data == "normal":
        kv.normal_()
    else:                       # "zeros": the same bytes moved, no switching in the arithmetic
        kv.zero_()
    q = torch.randn(B, Hq, D, device=dev).to(torch.bfloat16)
    if outliers:       # short sequences own ceil(ctx / block) blocks, the long ones a full row
        perm = torch.randperm(nblk).to(torch.int32)
        bt_h = torch.zeros(B, bps, dtype=torch.int32)
        short = (ctx + BLOCK - 1) // BLOCK
        pos = 0
        for i in range(B):
            n = bps if i < len(outliers) else short
            bt_h[i, :n] = perm[pos:pos + n]
            pos += n
        bt = bt_h.to(dev)
    else:
        bt = torch.randperm(nblk).to(torch.int32).view(B, bps).to(dev)
    if ragged:
        lens = torch.randint(1, ctx + 1, (B,), dtype=torch.int32, generator=torch.Generator().manual_seed(0))
    else:
        lens = torch.full((B,), ctx, dtype=torch.int32)
    for i, n in enumerate(outliers):
        lens[i] = n
    ctx_t = lens.to(dev)
    out = torch.empty_like(q)
    # launch order: longest first, as ModelRunner.prepare_decode passes it
    order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev) if ordered else None
    dplan = None
    if plan is not None:
        from light_vllm_amd.decode_plan import build_plan
        dplan = build_plan(lens.tolist(), Hkv, Hq // Hkv, **plan).to(dev)
    state = {"i": 0}

    def fn():
        l = state["i"] % layers
        state["i"] += 1
        ops.decode_attention(q, kv[l, 0], kv[l, 1], ctx_t, bt, D ** -0.5, num_splits=nsplit, out=out, seq_order=order,
                             plan=dplan)

    if graph:
        ops.reserve_decode_workspace(dev, 256 << 20)
    t = timeit_graph(fn) if graph else timeit(fn)
    tot = int(lens.sum())
    nb = int(((lens + BLOCK - 1) // BLOCK).sum())
    bytes_ = tot * Hkv * D * 4 + 2 * B * Hq * D * 2 + 4 * B + 4 * nb
    ns = nsplit or ops.decode_num_splits(B, Hkv, bps, BLOCK)
    return dict(kind="decode", B=B, ctx=ctx, Hq=Hq, Hkv=Hkv, nsplit=ns, ragged=ragged, ordered=ordered, data=data,
                kernel=ops.DEFAULT_KERNEL["decode"], us=round(t * 1e6, 1),
                GBs=round(bytes_ / t / 1e9, 1), pct_peak=round(bytes_ / t / 8e12 * 100, 1))


def prefill_case(lens, Hq, Hkv, paged_prefix=0, data="normal", iters=50, kernel=None, num_splits=0, order=None):
    """order: falsy = the library's own tile mapping; else a host-built prefill_order table."""
    total = sum(lens)
    tile_order = None
    if order:
        tile_order = ops.prefill_tile_order(lens, lens, Hq, Hkv, dev, num_splits=num_splits, kernel=kernel, force=True)

    def draw(h):
        if data == "uniform":      # uniform in [-2, 2), as tools/prefill_diag.hip
            return (torch.rand(total, h, D, device=dev) * 4 - 2).to(torch.bfloat16)
        if data == "zeros":
            return torch.zeros(total, h, D, dtype=torch.bfloat16, device=dev)
        return torch.randn(total, h, D, device=dev).to(torch.bfloat16)
    q, k, v = draw(Hq), draw(Hkv), draw(Hkv)
    cu = torch.tensor([0] + torch.tensor(lens).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    out = torch.empty_like(q)

    def fn():
        ops.prefill_attention(q, k, v, cu, cu, max(lens), max(lens), D ** -0.5, out=out, kernel=kernel, num_splits=num_splits,
                              tile_order=tile_order)

    # matrix-bound kernels run ~19 % slower for the first ~10 ms after an idle GPU (clock ramp):
    # at least 20 untimed launches and 25 ms of them first, so that the timed ones see the clocks a
    # busy engine has
    import time
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_warm = 0
    while n_warm < 20 or time.perf_counter() - t0 < 0.025:
        fn()
        n_warm += 1
        if n_warm % 10 == 0:
            torch.cuda.synchronize()
    t = timeit(fn, iters=iters, warmup=2)
    flops = sum(4 * Hq * D * (L * L - L * (L - 1) // 2) for L in lens)
    return dict(kind="prefill", data=data, iters=iters, lens=lens if len(lens) <= 8 else f"{len(lens)} seqs", Hq=Hq, Hkv=Hkv,
                kernel=kernel or ops.DEFAULT_KERNEL["prefill"], runs=num_splits, order=order,
                ms=round(t * 1e3, 3), TFLOPs=round(flops / t / 1e12, 1), pct_peak=round(flops / t / 2.5e15 * 100, 1))


def sdpa_case(B, L, Hq, Hkv, iters=50):
    """The platform library on the same job, for scale only (nothing here ships it): torch's
    scaled_dot_product_attention, causal, B equal prompts of L tokens, kv heads shared by enable_gqa."""
    import torch.nn.functional as F
    q = torch.randn(B, Hq, L, D, device=dev).to(torch.bfloat16)
    k = torch.randn(B, Hkv, L, D, device=dev).to(torch.bfloat16)
    v = torch.randn(B, Hkv, L, D, device=dev).to(torch.bfloat16)
    out = {}
    for name, backend in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION),
                          ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
        def fn():
            with torch.nn.attention.sdpa_kernel(backend):
                return F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=(Hq != Hkv))
        try:
            for _ in range(25):
                fn()
            torch.cuda.synchronize()
            t = timeit(fn, iters=iters, warmup=2)
        except Exception as e:  # noqa: BLE001 - a backend the build lacks is a result, not a failure
            out[name] = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
            continue
        flops = B * 4 * Hq * D * (L * L - L * (L - 1) // 2)
        out[name] = dict(ms=round(t * 1e3, 3), TFLOPs=round(flops / t / 1e12, 1))
    return dict(kind="torch_sdpa", B=B, L=L, Hq=Hq, Hkv=Hkv, backends=out)


def prefix_case(B, new, cached, Hq, Hkv, nsplit):
    """Prefix-cached prefill: `new` query tokens per sequence on top of `cached` tokens in the paged
    cache (Lk = cached + new), nsplit KV runs per query tile (0 = library heuristic, 1 = one pass)."""
    Lk = cached + new
    bps = (Lk + BLOCK - 1) // BLOCK
    nblk = B * bps
    kc = torch.randn(nblk, BLOCK, Hkv, D, device=dev).to(torch.bfloat16)
    vc = torch.randn(nblk, BLOCK, Hkv, D, device=dev).to(torch.bfloat16)
    q = torch.randn(B * new, Hq, D, device=dev).to(torch.bfloat16)
    bt = torch.randperm(nblk).to(torch.int32).view(B, bps).to(dev)
    cu_q = torch.arange(0, (B + 1) * new, new, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * Lk, Lk, dtype=torch.int32, device=dev)
    out = torch.empty_like(q)

    def fn():
        ops.prefill_attention(q, kc, vc, cu_q, cu_k, new, Lk, D ** -0.5, block_table=bt, out=out, num_splits=nsplit)

    t = timeit_graph(fn, iters=10, warmup=5)
    flops = B * 4 * Hq * D * (new * Lk - new * (new - 1) // 2)
    ns = nsplit or ops._lib.load().lv_prefill_num_splits(B, new, Lk, Hq)
    return dict(kind="prefill_prefix", B=B, new=new, cached=cached, Hq=Hq, Hkv=Hkv, nsplit=ns, us=round(t * 1e6, 1),
                TFLOPs=round(flops / t / 1e12, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="decode")
    ap.add_argument("--lib", default=None, help="another build of liblvattn.so to measure (A/B runs)")
    ap.add_argument("--decode-kernel", default="auto", choices=list(ops.DECODE_KERNELS))
    ap.add_argument("--prefill-kernel", default="auto", choices=list(ops.PREFILL_KERNELS))
    args = ap.parse_args()
    ops.DEFAULT_KERNEL.update(decode=args.decode_kernel, prefill=args.prefill_kernel)
    if args.lib:
        from light_vllm_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)
    rows = []
    if args.what in ("decode", "all"):
        for (Hq, Hkv) in [(16, 8), (32, 8), (64, 8)]:
            rows.append(decode_case(256, 1024, Hq, Hkv))
        rows.append(decode_case(256, 1024, 32, 8, ragged=True))
        rows.append(decode_case(256, 1024, 32, 8, ragged=True, ordered=True))
        rows.append(decode_case(256, 1024, 16, 8, ragged=True))
        rows.append(decode_case(256, 1024, 16, 8, ragged=True, ordered=True))
        rows.append(decode_case(64, 4096, 32, 8))
        for (Hq, Hkv) in [(16, 4), (8, 2), (4, 1), (8, 1)]:
            for B in (1, 8, 64):
                rows.append(decode_case(B, 32768, Hq, Hkv))
        rows.append(decode_case(1, 32768, 32, 8))
        rows.append(decode_case(8, 1024, 32, 8))
        rows.append(decode_case(1, 1024, 32, 8))
    if args.what in ("prefill", "all"):
        rows.append(prefill_case([4096] * 4, 32, 8))
        rows.append(prefill_case([4096, 3000, 1024, 8264], 32, 8))
        rows.append(prefill_case([1024] * 16, 16, 8))
        rows.append(prefill_case([4096] * 4, 8, 1))
    if args.what == "prefill_sdpa":      # this library beside torch's own attention on the bench's prefill jobs
        for (Hq, Hkv) in ((32, 8), (8, 1)):
            rows.append(prefill_case([4096] * 4, Hq, Hkv))
            rows.append(sdpa_case(4, 4096, Hq, Hkv))
    if args.what == "prefill_data":      # same launch, different operand values and burst lengths
        for data in ("normal", "uniform", "zeros"):
            for iters in (10, 100):
                rows.append(prefill_case([4096] * 4, 32, 8, data=data, iters=iters))
    if args.what == "prefill_shapes":    # the shapes the kernel choice is made on
        for lens, Hq, Hkv in (([512] * 32, 32, 8), ([1024] * 16, 32, 8), ([1536] * 10, 32, 8), ([2048] * 8, 32, 8),
                              ([4096] * 4, 32, 8), ([1024] * 16, 16, 8), ([2048] * 8, 4, 1)):
            rows.append(prefill_case(lens, Hq, Hkv))
    if args.what == "prefill_len":       # 16384 tokens cut into fewer, longer prompts: the cost per query tile beside the cost per key tile
        for kernel in ("pp", "w8", "w64"):
            for L in (2048, 4096, 8192, 16384):
                rows.append(prefill_case([L] * (16384 // L), 32, 8, kernel=kernel))
    if args.what == "prefill_persist1":
        for order in (None, "desc", None, "desc"):
            rows.append(prefill_case([4096] * 4, 32, 8, order=order))
        for order in (None, "desc"):
            rows.append(prefill_case([8192] * 2, 32, 8, order=order))
    if args.what == "prefill_persist":   # one query tile per workgroup against the persistent form (item tables)
        for lens, Hq, Hkv in (([4096] * 4, 32, 8), ([2048] * 8, 32, 8), ([8192] * 2, 32, 8), ([4096, 3000, 1024, 8264], 32, 8),
                              ([4096] * 4, 8, 1), ([4096] * 4, 16, 4)):
            for order in (None, "desc", None, "desc"):
                rows.append(prefill_case(lens, Hq, Hkv, order=order))
    if args.what == "prefill_tp8":       # one rank's share of a 4 x 4096 prefill at TP = 8: kernels x KV runs
        ops.reserve_decode_workspace(dev, 1 << 30)
        for Hq, Hkv in ((8, 1), (4, 1)):
            for kernel in ("pp", "w8", "w64", "w4"):
                rows.append(prefill_case([4096] * 4, Hq, Hkv, kernel=kernel, num_splits=1))
            for ns in (0, 2, 4):
                rows.append(prefill_case([4096] * 4, Hq, Hkv, kernel="w8", num_splits=ns))
    if args.what == "prefill_split_full":  # whole prompts (no cached prefix) on small grids: one pass vs forced KV runs
        ops.reserve_decode_workspace(dev, 1 << 30)
        for B, L, Hq, Hkv in ((1, 4096, 4, 1), (2, 4096, 4, 1), (4, 4096, 4, 1), (1, 4096, 8, 1), (2, 4096, 8, 1),
                              (1, 8192, 4, 1), (1, 16384, 4, 1), (1, 2048, 4, 1), (2, 2048, 8, 1), (1, 4096, 32, 8)):
            for ns in (1, 2, 4, 8):
                rows.append(prefill_case([L] * B, Hq, Hkv, num_splits=ns, iters=30))
    if args.what in ("prefill_burn", "decode_burn"):   # one launch shape for ~6 s, for tools/power_probe.sh to sample beside
        import time as _time
        if args.what == "prefill_burn":
            total = 4 * 4096
            q = torch.randn(total, 32, D, device=dev).to(torch.bfloat16)
            k = torch.randn(total, 8, D, device=dev).to(torch.bfloat16)
            v = torch.randn(total, 8, D, device=dev).to(torch.bfloat16)
            cu = torch.tensor([0, 4096, 8192, 12288, 16384], dtype=torch.int32, device=dev)
            out = torch.empty_like(q)
            fn = lambda: ops.prefill_attention(q, k, v, cu, cu, 4096, 4096, D ** -0.5, out=out)  # noqa: E731
            unit = 4 * 4 * 32 * D * (4096 * 4096 - 4096 * 4095 // 2) / 1e12      # TFLOP per launch
        else:
            nblk = 256 * 4
            kv = torch.empty(4, 2, nblk, BLOCK, 8, D, dtype=torch.bfloat16, device=dev).normal_()
            q = torch.randn(256, 32, D, device=dev).to(torch.bfloat16)
            bt = torch.randperm(nblk).to(torch.int32).view(256, 4).to(dev)
            ctx_t = torch.full((256,), 1024, dtype=torch.int32, device=dev)
            out = torch.empty_like(q)
            st = {"i": 0}

            def fn():
                st["i"] += 1
                ops.decode_attention(q, kv[st["i"] % 4, 0], kv[st["i"] % 4, 1], ctx_t, bt, D ** -0.5, out=out)
            unit = 256 * 1024 * 8 * D * 4 / 1e12                                   # TB per launch
        t_end = _time.perf_counter() + 6.0
        n = 0
        t0 = _time.perf_counter()
        while _time.perf_counter() < t_end:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        dt = _time.perf_counter() - t0
        rows.append(dict(kind=args.what, launches=n, seconds=round(dt, 2), rate=round(n * unit / dt, 1),
                         unit="TFLOP/s" if args.what == "prefill_burn" else "TB/s"))
    if args.what == "prefill_small":     # grids that fit the chip at once (TP = 8 shapes)
        for lens, Hq, Hkv in (([2048] * 8, 4, 1), ([4096] * 4, 4, 1), ([1024] * 16, 4, 1), ([4096] * 2, 8, 2),
                              ([1024] * 16, 32, 8), ([4096] * 4, 32, 8)):
            rows.append(prefill_case(lens, Hq, Hkv))
    if args.what == "prefill_prefix":    # few new tokens on a long cached prefix: one pass vs KV runs
        for (B, new, cached, Hq, Hkv) in [(8, 100, 4096, 8, 1), (8, 100, 4096, 32, 8), (1, 100, 32000, 8, 1),
                                          (4, 300, 1024, 8, 1), (2, 512, 8192, 32, 8), (1, 4096, 0, 4, 1),
                                          (6, 300, 1024, 8, 1)]:
            for ns in (1, 0, 2, 4, 8):
                rows.append(prefix_case(B, new, cached, Hq, Hkv, ns))
    if args.what == "prefill_order":     # library mapping vs host-built tile tables
        for lens, Hq, Hkv in (([4096] * 4, 8, 1), ([4096] * 4, 4, 1), ([4096] * 2, 8, 2), ([4096] * 4, 16, 4),
                              ([2048] * 8, 4, 1), ([4096, 3000, 1024, 8264], 8, 1), ([4096, 3000, 1024, 8264], 32, 8),
                              ([4096] * 4, 32, 8), ([1024] * 16, 4, 1), ([8192] * 2, 8, 1)):
            for order in (None, "desc"):
                rows.append(prefill_case(lens, Hq, Hkv, order=order))
    if args.what == "prefill_l2order":   # 8 kv heads: which sequences an XCD works on at the same time (L2 working set)
        from light_vllm_amd.prefill_order import TileOrder
        lens, qt = [4096] * 4, 16
        tables = {
            "library": None,
            "heaviest tile of every sequence first": [b * qt + t for t in range(qt - 1, -1, -1) for b in range(4)],
            "one sequence after another": [b * qt + t for b in range(4) for t in range(qt - 1, -1, -1)],
            "two sequences at a time": [b * qt + t for pair in ((0, 1), (2, 3)) for t in range(qt - 1, -1, -1) for b in pair],
        }
        for name, codes in tables.items():
            for _ in range(2):
                r = prefill_case(lens, 32, 8)
                if codes is not None:
                    order = TileOrder(codes, 256, 1, 4096).to(dev)
                    q = torch.randn(sum(lens), 32, D, device=dev).to(torch.bfloat16)
                    k = torch.randn(sum(lens), 8, D, device=dev).to(torch.bfloat16)
                    v = torch.randn(sum(lens), 8, D, device=dev).to(torch.bfloat16)
                    cu = torch.tensor([0, 4096, 8192, 12288, 16384], dtype=torch.int32, device=dev)
                    out = torch.empty_like(q)
                    fn = lambda: ops.prefill_attention(q, k, v, cu, cu, 4096, 4096, D ** -0.5, out=out, tile_order=order)
                    for _ in range(30):
                        fn()
                    torch.cuda.synchronize()
                    t = timeit(fn, iters=50, warmup=2)
                    r = dict(kind="prefill", ms=round(t * 1e3, 3), dropped=ops.STATS["tile_order_dropped"])
                r["order"] = name
                rows.append(r)
    if args.what == "prefill_order2":    # small grids: kernels x KV runs under a host-built table
        ops.reserve_decode_workspace(dev, 1 << 30)
        for lens, Hq, Hkv in (([4096] * 4, 4, 1), ([4096] * 2, 8, 2), ([4096] * 4, 8, 1), ([2048] * 8, 4, 1)):
            for kernel, ns in (("auto", 0), ("w4", 1), ("w8", 1), ("w8", 2), ("w4", 2), ("pp", 1), ("w64", 1)):
                for order in (None, "desc"):
                    rows.append(prefill_case(lens, Hq, Hkv, kernel=kernel, num_splits=ns, order=order))
    if args.what == "prefill1":
        rows.append(prefill_case([4096] * 4, 32, 8))
    if args.what == "decode_c2":         # the bench launch: both kernels, random and all-zero caches, eager and replayed
        for data in ("normal", "zeros"):
            for graph in (False, True):
                rows.append(decode_case(256, 1024, 16, 8, data=data, graph=graph))
                rows.append(decode_case(256, 1024, 32, 8, data=data, graph=graph))
    if args.what == "decode_drift":      # the same two launches again and again: does the chip's state drift?
        import time as _time
        t_start = _time.perf_counter()
        for rep in range(12):
            for (Hq, Hkv) in ((16, 8), (32, 8)):
                r = decode_case(256, 1024, Hq, Hkv, layers=4, graph=True)
                r["t_s"] = round(_time.perf_counter() - t_start, 1)
                rows.append(r)
    if args.what == "decode_footprint":  # the same launch rotating over 4, 12 and 36 layer caches (8, 24, 72 GiB)
        for layers in (4, 12, 36, 4):
            for (Hq, Hkv) in ((16, 8), (32, 8)):
                r = decode_case(256, 1024, Hq, Hkv, layers=layers, graph=True)
                r["layers"] = layers
                rows.append(r)
    if args.what == "decode_8b":         # the north-star shapes and their TP shards, uniform and ragged
        for (Hq, Hkv) in [(32, 8), (16, 4), (8, 2), (4, 1), (64, 8), (8, 1)]:
            rows.append(decode_case(256, 1024, Hq, Hkv, graph=True))
        rows.append(decode_case(256, 1024, 32, 8, ragged=True, ordered=True, graph=True))
        rows.append(decode_case(256, 4096, 32, 8, ragged=True, ordered=True, graph=True))
        rows.append(decode_case(32, 32768, 32, 8, graph=True))
        rows.append(decode_case(32, 32768, 4, 1, graph=True))
        rows.append(decode_case(64, 4096, 32, 8, graph=True))
    if args.what == "decode_c5":         # Qwen3-32B TP=8 shard at 32k context: eager vs replayed, split counts
        for graph in (False, True):
            for ns in (0, 1, 2, 4, 8):
                rows.append(decode_case(64, 32768, 8, 1, nsplit=ns, graph=graph))
    if args.what == "decode_c4_tp":      # BASELINE config 4 (batch 32, 32k context) per rank at TP = 1, 2, 4, 8
        for (Hq, Hkv) in [(32, 8), (16, 4), (8, 2), (4, 1)]:
            rows.append(decode_case(32, 32768, Hq, Hkv, graph=True))
    if args.what == "decode_g8":
        rows.append(decode_case(256, 1024, 64, 8))
    if args.what == "decode_ragged1":    # ragged batches, one pass, graph replay: for comparing builds
        for (Hq, Hkv) in [(32, 8), (16, 8), (64, 8)]:
            for ctx in (1024, 4096):
                rows.append(decode_case(256, ctx, Hq, Hkv, nsplit=1, ragged=True, ordered=True, graph=True))
    if args.what == "decode_plan":       # ragged batches: longest-first order against host-built work lists
        for (Hq, Hkv) in [(32, 8), (16, 8), (64, 8)]:
            for ctx in (1024, 4096):
                rows.append(decode_case(256, ctx, Hq, Hkv, nsplit=1, ragged=True, ordered=True, graph=True))
                for kw in (dict(piece_tokens=256), dict(piece_tokens=384), dict(piece_tokens=512), dict(piece_tokens=1024), dict()):
                    r = decode_case(256, ctx, Hq, Hkv, ragged=True, graph=True, plan=kw)
                    r["plan"] = kw or "auto"
                    rows.append(r)
    if args.what == "decode_outliers":   # a few long contexts among short ones: one split count for all against a work list
        for (Hq, Hkv) in [(32, 8), (16, 8)]:
            for outl in ((16384,) * 4, (32768,), (8192,) * 8):
                for ns in (1, 0):
                    rows.append(decode_case(256, 1024, Hq, Hkv, nsplit=ns, ragged=True, ordered=True, graph=True, outliers=outl))
                for kw in (dict(piece_tokens=512), dict(piece_tokens=1024), dict(piece_tokens=2048), dict()):
                    r = decode_case(256, 1024, Hq, Hkv, ragged=True, graph=True, plan=kw, outliers=outl)
                    r["plan"] = kw or "auto"
                    rows.append(r)
                rows[-1]["outliers"] = list(outl)
    if args.what == "decode_ragged":
        for (Hq, Hkv) in [(32, 8), (16, 8)]:
            for ns in (1, 2, 4, 8):
                rows.append(decode_case(256, 1024, Hq, Hkv, nsplit=ns, ragged=True, ordered=True))
            rows.append(decode_case(256, 4096, Hq, Hkv, nsplit=1, ragged=True, ordered=True))
            rows.append(decode_case(256, 4096, Hq, Hkv, nsplit=4, ragged=True, ordered=True))
            rows.append(decode_case(256, 4096, Hq, Hkv, nsplit=16, ragged=True, ordered=True))
    if args.what == "decode_ctx":
        for ctx in (128, 256, 512, 1024, 2048):
            rows.append(decode_case(256, ctx, 32, 8))
            rows.append(decode_case(256, ctx, 16, 8))
    if args.what == "decode_small":      # hipGraph replay: the GPU time of launch-bound shapes
        for (Hq, Hkv) in [(32, 8), (16, 4), (4, 1)]:
            for (B, ctx) in [(1, 1024), (8, 1024), (1, 32768), (8, 32768)]:
                for ns in (0, 1):
                    rows.append(decode_case(B, ctx, Hq, Hkv, nsplit=ns, graph=True))
    if args.what == "fused":             # elementwise kernels either side of attention, prefill sizes
        N, E, I, Hq, Hkv = 16384, 4096, 12288, 32, 8
        x = torch.randn(N, 2 * I, device=dev).to(torch.bfloat16)
        o = torch.empty(N, I, dtype=torch.bfloat16, device=dev)
        t = timeit(lambda: ops.silu_and_mul(x, out=o))
        rows.append(dict(kind="silu_and_mul", N=N, I=I, us=round(t * 1e6, 1), GBs=round(N * I * 6 / t / 1e9, 1)))
        h = torch.randn(N, E, device=dev).to(torch.bfloat16)
        r = torch.randn(N, E, device=dev).to(torch.bfloat16)
        w = torch.randn(E, device=dev).to(torch.bfloat16)
        t = timeit(lambda: ops.add_rms_norm(h, r, w, 1e-6))
        rows.append(dict(kind="add_rms_norm", N=N, E=E, us=round(t * 1e6, 1), GBs=round(N * E * 8 / t / 1e9, 1)))
        del x, o
        qkv = torch.randn(N, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
        q, k, v = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
        pos = torch.arange(N, device=dev) % 4096
        qw = torch.randn(D, device=dev).to(torch.bfloat16)
        cs = torch.randn(4096, D, device=dev)
        nblk = N // BLOCK
        kc = torch.zeros(nblk, BLOCK, Hkv, D, dtype=torch.bfloat16, device=dev)
        vc = torch.zeros_like(kc)
        slots = torch.randperm(N, device=dev).to(torch.int32)
        t = timeit(lambda: ops.qk_norm_rope_store(q.view(N, Hq, D), k.view(N, Hkv, D), v.view(N, Hkv, D), pos, qw, qw, 1e-6,
                                                  cs, kc, vc, slots))
        nbytes = N * D * 2 * (2 * Hq + 2 * Hkv + Hkv + 2 * Hkv)
        rows.append(dict(kind="qk_norm_rope_store", N=N, Hq=Hq, Hkv=Hkv, us=round(t * 1e6, 1), GBs=round(nbytes / t / 1e9, 1)))
    if args.what == "decode_b1":         # one long sequence: split pass vs merge pass (profile this one)
        rows.append(decode_case(1, 32768, int(os.environ.get("KB_HQ", "4")), int(os.environ.get("KB_HKV", "1")),
                                nsplit=int(os.environ.get("KB_NS", "0")), graph=True))
    if args.what == "decode_mid":        # launches around the point where splitting stops paying
        for (B, ctx, Hq, Hkv) in [(64, 4096, 32, 8), (32, 4096, 32, 8), (16, 4096, 32, 8), (128, 4096, 4, 1), (64, 32768, 16, 4)]:
            for ns in (0, 1, 2, 4):
                rows.append(decode_case(B, ctx, Hq, Hkv, nsplit=ns, graph=True))
    if args.what == "decode_tp":         # per-rank shapes of Qwen3-8B at TP = 2 / 4 / 8, batch 256
        for (Hq, Hkv) in [(16, 4), (8, 2), (4, 1)]:
            for ctx in (1024, 4096):
                for ns in (0, 1, 2, 4, 8):
                    rows.append(decode_case(256, ctx, Hq, Hkv, nsplit=ns, graph=True))
    if args.what == "decode_splits":     # split-count sweep around the heuristic, hipGraph replay
        for (Hq, Hkv) in [(4, 1), (16, 4), (32, 8)]:
            for (B, ctx) in [(1, 32768), (8, 32768), (1, 4096), (8, 4096), (32, 4096), (1, 1024), (8, 1024)]:
                for ns in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
                    if ns and (ns > ctx // 64 or ctx // ns > 4096):
                        continue
                    rows.append(decode_case(B, ctx, Hq, Hkv, nsplit=ns, graph=True))
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

"""Probe: does the time of one fixed decode launch (256 x 1024 tokens, 16q/8kv and 32q/8kv, 4 rotating
layer caches) depend on what the process did just before? It does: for up to ~0.8 s after a hipFree of
a GiB or more, or a large fill, the launch runs 3-5 % slower on the same untouched cache, then returns
to the time it holds over thousands of back-to-back launches. Prints one JSON line per stage."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from light_vllm_amd import ops  # noqa: E402

D, BLOCK, B, CTX, HKV, LAYERS = 128, 256, 256, 1024, 8, 4
dev = torch.device("cuda:0")


class Case:
    def __init__(self):
        bps = CTX // BLOCK
        nblk = B * bps
        self.kv = torch.empty(LAYERS, 2, nblk, BLOCK, HKV, D, dtype=torch.bfloat16, device=dev).normal_()
        self.bt = torch.randperm(nblk, generator=torch.Generator().manual_seed(1)).to(torch.int32).view(B, bps).to(dev)
        self.ctx = torch.full((B,), CTX, dtype=torch.int32, device=dev)
        self.q = {hq: torch.randn(B, hq, D, device=dev).to(torch.bfloat16) for hq in (16, 32)}
        self.out = {hq: torch.empty_like(q) for hq, q in self.q.items()}

    def time(self, hq, iters=20):
        i = [0]

        def fn():
            l = i[0] % LAYERS
            i[0] += 1
            ops.decode_attention(self.q[hq], self.kv[l, 0], self.kv[l, 1], self.ctx, self.bt, D ** -0.5, out=self.out[hq])

        for _ in range(4):
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
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (5 * iters) * 1e3


def report(tag, case):
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({"stage": tag, "us_16q": round(case.time(16), 1), "us_32q": round(case.time(32), 1),
                      "kv_ptr": hex(case.kv.data_ptr()), "device_used_GiB": round((total - free) / 2**30, 1)}), flush=True)


import time


def series(case, hq, chunks=25, iters=20, replays=5):
    """Launch time per chunk of iters * replays back-to-back launches, no host pause between chunks."""
    i = [0]

    def fn():
        l = i[0] % LAYERS
        i[0] += 1
        ops.decode_attention(case.q[hq], case.kv[l, 0], case.kv[l, 1], case.ctx, case.bt, D ** -0.5, out=case.out[hq])

    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    torch.cuda.synchronize()
    time.sleep(2.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    ev[0].record()
    for c in range(chunks):
        for _ in range(replays):
            g.replay()
        ev[c + 1].record()
    torch.cuda.synchronize()
    return [round(ev[c].elapsed_time(ev[c + 1]) / (iters * replays) * 1e3, 1) for c in range(chunks)]


a = Case()
for hq in (16, 32):
    print(json.dumps({"stage": f"{hq}q: 25 chunks of 100 launches from 2 s of idle (us per launch)", "series": series(a, hq)}), flush=True)
# the same after a burst of other work: fill 16 GiB with normal numbers, free it
big = torch.empty(8 << 30, dtype=torch.bfloat16, device=dev).normal_()
del big
torch.cuda.empty_cache()
report("right after a 16 GiB normal_() fill and its hipFree", a)
time.sleep(1.0)
report("1 s later", a)
for gib in (0.0625, 1, 4):
    big = torch.empty(int(gib * 2**30), dtype=torch.uint8, device=dev)
    del big
    torch.cuda.empty_cache()
    report(f"right after hipMalloc + hipFree of {gib} GiB (never touched)", a)
    time.sleep(1.0)
    report("1 s later", a)

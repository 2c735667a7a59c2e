"""Probe: does this stack's RCCL (torch.distributed backend "nccl") initialise, reduce, gather objects
and let an all-reduce be captured into a hipGraph? One rank on one GPU: the degenerate group, but the
same API path bench.py --gpus N and the engine take when the direct all-reduce is unavailable."""
import json
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
out = {}
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
out["backend"] = dist.get_backend()
x = torch.ones(32, 4096, dtype=torch.bfloat16, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
out["eager_all_reduce"] = float(x.float().sum()) == 32 * 4096
got = [None]
dist.all_gather_object(got, ("ok", b"\x00" * 64))
out["all_gather_object"] = got[0][0] == "ok"
dist.barrier()
try:
    g = torch.cuda.CUDAGraph()
    y = torch.ones(32, 4096, dtype=torch.bfloat16, device=dev)
    dist.all_reduce(y)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        dist.all_reduce(y)
    y.fill_(2.0)
    g.replay()
    torch.cuda.synchronize()
    out["graph_capture_all_reduce"] = float(y.float().mean()) == 2.0
except Exception as exc:   # noqa: BLE001
    out["graph_capture_all_reduce"] = repr(exc)[:200]
parts = [torch.empty_like(x)]
dist.all_gather(parts, x)
out["all_gather"] = bool(torch.equal(parts[0], x))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))

"""Test-side carrier for GPU tensors over a host-only ("gloo") process group: the collectives of
ranks that share ONE GPU, where RCCL cannot run (it refuses two ranks on one device). Installed
with layers.tp.set_carrier by the single-GPU rehearsals of the tensor-parallel path; it carries
what the direct IPC all-reduce does not take (prefill-sized all-reduces, the logits gather)."""
import torch
import torch.distributed as dist


class HostStagedCarrier:

    def all_reduce(self, t):
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)
        return t

    def all_gather(self, t):
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, h)
        return torch.stack(parts).to(t.device)

    def gather_to_rank0(self, t):
        rank0 = dist.get_rank() == 0
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(dist.get_world_size())] if rank0 else None
        dist.gather(h, parts, 0)
        return [p.to(t.device) for p in parts] if rank0 else None

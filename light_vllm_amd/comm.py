"""Direct small-message all-reduce between the ranks of one node (include/lv_comm.h): the
transport the tensor-parallel layers use for decode-sized sums instead of torch.distributed
(reference call sites: dist.all_reduce at lightvllm/layers/linear.py:388-389 and
layers/embed_head.py:112). One communicator per process, created after the process group exists;
the HIP IPC handles of the ranks' buffers travel through that group (all_gather_object), the
data never does.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib

MAX_BYTES = 8 << 20          # LV_COMM_MAX_BYTES
HANDLE_BYTES = 64            # LV_COMM_HANDLE_BYTES
ALGORITHMS = {"auto": 0, "one_shot": 1, "two_shot": 2}


class DirectAllReduce:
    """This rank's end of an lv_comm communicator on `device`. Construction is local (allocate the
    buffers, export the handle); `connect` maps the peers; the exchange of the handles in between
    is the caller's (create_for_group does it over a torch.distributed group)."""

    def __init__(self, device: torch.device, rank: int, world_size: int, timeout_ms: int = 0):
        self.device = torch.device(device)
        self.rank, self.world_size = rank, world_size
        self._lib = _lib.load()
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lv_comm_create(rank, world_size, timeout_ms, ctypes.byref(self._comm)), "lv_comm_create")
            mine = ctypes.create_string_buffer(HANDLE_BYTES)
            _lib.check(self._lib.lv_comm_get_handle(self._comm, mine), "lv_comm_get_handle")
        self.handle = bytes(mine.raw)
        self._inflow = int(self._lib.lv_comm_one_shot_inflow(self._comm))     # cached: usable_fused runs per layer

    def connect(self, handles) -> None:
        """handles: every rank's `handle` in rank order."""
        assert len(handles) == self.world_size and all(isinstance(h, bytes) and len(h) == HANDLE_BYTES for h in handles)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.lv_comm_connect(self._comm, b"".join(handles)), "lv_comm_connect")

    @property
    def one_shot_inflow(self) -> int:
        """"auto" goes two shot once (world_size - 1) * bytes passes this (lv_comm_one_shot_inflow)."""
        return int(self._lib.lv_comm_one_shot_inflow(self._comm)) if self._comm is not None else 0

    def set_one_shot_inflow(self, nbytes: int) -> None:
        """The same value on every rank (create_for_group takes rank 0's)."""
        _lib.check(self._lib.lv_comm_set_one_shot_inflow(self._comm, int(nbytes)), "lv_comm_set_one_shot_inflow")
        self._inflow = int(nbytes)

    def algorithm_for(self, nbytes: int) -> str:
        return "two_shot" if (self.world_size - 1) * nbytes > self.one_shot_inflow else "one_shot"

    def usable(self, t: torch.Tensor) -> bool:
        nbytes = t.numel() * t.element_size()
        return (self._comm is not None and t.is_cuda and t.device == self.device and t.dtype == torch.bfloat16
                and t.is_contiguous() and 0 < nbytes <= MAX_BYTES and nbytes % 16 == 0 and t.data_ptr() % 16 == 0)

    def all_reduce(self, t: torch.Tensor, algorithm: str = "auto") -> torch.Tensor:
        """In-place bf16 sum over the ranks (fp32 accumulation in rank order, identical bits on
        every rank). Stream-ordered on torch's current stream; capturable into a hipGraph.
        algorithm: "auto", "one_shot" or "two_shot" (same result; the same on every rank)."""
        assert self.usable(t), "direct all-reduce: contiguous bf16 GPU tensor of at most 8 MiB, 16-byte multiple"
        with torch.cuda.device(self.device):
            rc = self._lib.lv_comm_all_reduce(self._comm, t.data_ptr(), t.numel(), ALGORITHMS[algorithm],
                                              torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "lv_comm_all_reduce")
        return t

    def usable_fused(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor) -> bool:
        """x, residual [rows, hidden] and weight [hidden] fit all_reduce_add_rms_norm, and the message
        is one the plain all-reduce would serve in one shot ((world - 1) * bytes <= one_shot_inflow)."""
        if not (self.usable(x) and x.dim() == 2 and residual.shape == x.shape and residual.dtype == x.dtype
                and residual.is_contiguous() and residual.device == x.device and residual.data_ptr() % 16 == 0):
            return False
        hidden = x.size(1)
        nbytes = x.numel() * 2
        return (weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.numel() == hidden
                and weight.device == x.device and weight.data_ptr() % 16 == 0
                and hidden % 8 == 0 and hidden <= 8192 and (self.world_size - 1) * nbytes <= self._inflow)

    def all_reduce_add_rms_norm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
        """x <- RMSNorm(all_reduce(x) + residual) * weight, residual <- all_reduce(x) + residual, both in
        place, one kernel; the same bits as all_reduce(x) followed by ops.add_rms_norm(x, residual, ...)."""
        assert self.usable_fused(x, residual, weight), "fused all-reduce + norm: [rows, hidden] bf16, one-shot sized"
        with torch.cuda.device(self.device):
            rc = self._lib.lv_comm_all_reduce_add_rms_norm(self._comm, x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                                           x.size(0), x.size(1), float(eps),
                                                           torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "lv_comm_all_reduce_add_rms_norm")
        return x, residual

    def failed(self) -> bool:
        return self._comm is None or bool(self._lib.lv_comm_failed(self._comm))

    def close(self) -> None:
        if self._comm is not None:
            with torch.cuda.device(self.device):
                self._lib.lv_comm_destroy(self._comm)
            self._comm = None


def _self_test(comm: DirectAllReduce) -> str:
    """Known-answer sums through both algorithms on the freshly connected communicator: rank r
    contributes (r + 1) * (i % 8) at element i, so every element of the result is the small integer
    world * (world + 1) / 2 * (i % 8), exact in bf16. Returns "" or what went wrong. A peer mapping
    that opened but does not carry data or flags between these GPUs shows up here (as a wrong sum
    or as the kernel's bounded wait running out), before the model depends on it."""
    world = comm.world_size
    for algorithm, n in (("one_shot", 4096), ("two_shot", 64 * 4096), ("auto", 1024 * 1024), ("one_shot", 8)):
        pattern = (torch.arange(n, device=comm.device) % 8).to(torch.float32)
        x = (pattern * (comm.rank + 1)).to(torch.bfloat16)
        comm.all_reduce(x, algorithm)
        torch.cuda.synchronize(comm.device)
        if comm.failed():
            return f"{algorithm} of {n} elements: a peer did not arrive within the timeout"
        want = (pattern * (world * (world + 1) // 2)).to(torch.bfloat16)
        if not torch.equal(x, want):
            bad = int((x != want).sum())
            return f"{algorithm} of {n} elements: {bad} wrong sums"
    return ""


def create_for_group(device: torch.device, timeout_ms: int = 0, group=None,
                     self_test: bool = True, one_shot_inflow_bytes: Optional[int] = None) -> Optional[DirectAllReduce]:
    """Collective over `group`: every rank builds its end, the handles travel through the group, every
    rank maps its peers, and (self_test) known-answer sums run through both algorithms. Each local
    step is followed by an exchange of its outcome, so that a rank on which a step fails (IPC can be
    unavailable, e.g. ranks in different containers) never leaves the others waiting in a different
    collective: all ranks keep the communicator or all drop it together and stay on
    torch.distributed. Returns None then.
    one_shot_inflow_bytes: the one-shot / two-shot threshold of "auto" (None: the library's default or
    LV_COMM_ONE_SHOT_INFLOW_BYTES); whatever rank 0 ends up with is set on every rank."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def agree(ok: bool, payload):
        """all-gather (ok, payload); returns (everyone ok, payloads in rank order)."""
        got = [None] * world
        dist.all_gather_object(got, (ok, payload), group=group)
        return all(g[0] for g in got), [g[1] for g in got]

    def give_up(comm, reasons):
        if comm is not None:
            comm.close()
        if rank == 0:
            print("direct all-reduce unavailable, staying on torch.distributed:",
                  "; ".join(f"rank {r}: {m}" for r, m in enumerate(reasons) if isinstance(m, str) and m), flush=True)
        return None

    comm, err = None, ""
    if 2 <= world <= 8:
        try:
            comm = DirectAllReduce(device, rank, world, timeout_ms)
        except Exception as exc:      # noqa: BLE001 - any failure means "not available on this rank"
            err = repr(exc)
    else:
        err = f"{world} ranks (2..8 supported)"
    if comm is not None and one_shot_inflow_bytes is not None:
        comm.set_one_shot_inflow(one_shot_inflow_bytes)
    ok, payloads = agree(comm is not None, (comm.handle, comm.one_shot_inflow) if comm is not None else err)
    if not ok:
        return give_up(comm, [p if isinstance(p, str) else "" for p in payloads])
    comm.set_one_shot_inflow(payloads[0][1])          # one threshold for the group: rank 0's
    payloads = [p[0] for p in payloads]
    err = ""
    try:
        comm.connect(payloads)
    except Exception as exc:          # noqa: BLE001
        err = repr(exc)
    ok, reasons = agree(not err, err)
    if not ok:
        return give_up(comm, reasons)
    dist.barrier(group=group)         # every rank has mapped every peer before the first use
    if self_test:
        try:
            err = _self_test(comm)
        except Exception as exc:      # noqa: BLE001
            err = repr(exc)
        ok, reasons = agree(not err, err)
        if not ok:
            dist.barrier(group=group)  # nobody unmaps while a peer's kernel may still read
            return give_up(comm, reasons)
    return comm


_ACTIVE: Optional[DirectAllReduce] = None


def install(comm: Optional[DirectAllReduce]) -> None:
    """Make `comm` the transport layers.tp.all_reduce uses for messages it can take (None: none)."""
    global _ACTIVE
    _ACTIVE = comm


def active() -> Optional[DirectAllReduce]:
    return _ACTIVE

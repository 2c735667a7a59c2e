"""Tensor-parallel linear layers (reference: lightvllm/layers/linear.py).

The GEMMs are plain F.linear (hipBLASLt on MI355X). What matters to the attention path is the
sharding: QKVParallelLinear gives rank r the query heads [r*Hq/tp, (r+1)*Hq/tp) and kv heads
[r*Hkv/tp, (r+1)*Hkv/tp) in a fused [q | k | v] output (linear.py:240-310), so attention needs no
communication, and RowParallelLinear sums the per-rank partial outputs with one in-place
all-reduce (linear.py:388-389) — RCCL over xGMI when the process group backend is "nccl".
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import tp
from .tp import divide, tp_rank, tp_size


class LinearBase(nn.Module):

    def __init__(self, input_size: int, output_size: int, tp_dim: int | None = None):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.tp_dim = tp_dim
        self.tp_rank = tp_rank()
        self.tp_size = tp_size()

    def _make(self, out_features: int, in_features: int, bias: bool, bias_size: int | None = None):
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(bias_size if bias_size is not None else out_features))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size)
        self._make(output_size, input_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear(x, self.weight, self.bias)


class ColumnParallelLinear(LinearBase):
    """Output features split across ranks; input replicated."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, 0)
        self.input_size_per_partition = input_size
        self.output_size_per_partition = divide(output_size, self.tp_size)
        self._make(self.output_size_per_partition, input_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear(x, self.weight, self.bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel projections fused in one GEMM (gate_proj | up_proj)."""

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias=bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(self.tp_dim, offset, size)
        src = loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank]
        assert dst.shape == src.shape
        dst.copy_(src)


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q/k/v projection; heads are the sharding unit."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads or total_num_heads
        n = tp_size()
        self.num_heads = divide(self.total_num_heads, n)
        self.num_kv_heads = divide(self.total_num_kv_heads, n)
        super().__init__(hidden_size, (self.total_num_heads + 2 * self.total_num_kv_heads) * head_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        assert loaded_shard_id in ("q", "k", "v")
        q = self.num_heads * self.head_size
        kv = self.num_kv_heads * self.head_size
        size, offset = {"q": (q, 0), "k": (kv, q), "v": (kv, q + kv)}[loaded_shard_id]
        dst = param.data.narrow(self.tp_dim, offset, size)
        src = loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank]
        assert dst.shape == src.shape
        dst.copy_(src)


class RowParallelLinear(LinearBase):
    """Input features split across ranks; partial outputs summed by all-reduce."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, 1)
        self.input_size_per_partition = divide(input_size, self.tp_size)
        self.output_size_per_partition = output_size
        self._make(output_size, self.input_size_per_partition, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        if param is self.bias:
            if self.tp_rank == 0:
                param.data.copy_(loaded_weight)
            return
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    def forward(self, x: torch.Tensor, reduce: bool = True) -> torch.Tensor:
        """reduce=False leaves this rank's partial sums: the caller reduces them together with the
        step that follows (layers.tp.all_reduce_add_rms_norm). Not with a bias, which is added once,
        after the sum."""
        y = F.linear(x, self.weight)
        if not reduce:
            assert self.bias is None
            return y
        if self.tp_size > 1:
            tp.all_reduce(y)
        if self.bias is not None:
            y = y + self.bias
        return y

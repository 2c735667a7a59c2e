"""Vocabulary-parallel embedding and LM head (reference: lightvllm/layers/embed_head.py).

Embedding: each rank holds vocab rows [r*V/tp, (r+1)*V/tp), looks up its own ids, zeroes the rest
and the ranks' partial rows are summed by all-reduce (:82-114). LM head: only the last token of
each prefill sequence is projected (:161-167); per-rank logits [B, V/tp] are gathered to rank 0 and
concatenated along the vocab axis (:175-195); other ranks return None.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from ..utils.context import get_context
from . import tp
from .tp import tp_rank, tp_size


class VocabParallelEmbedding(nn.Module):

    def __init__(self, num_embeddings: int, embedding_dim: int):
        super().__init__()
        self.tp_rank = tp_rank()
        self.tp_size = tp_size()
        assert num_embeddings % self.tp_size == 0
        self.num_embeddings = num_embeddings
        self.num_embeddings_per_partition = num_embeddings // self.tp_size
        self.vocab_start_idx = self.num_embeddings_per_partition * self.tp_rank
        self.vocab_end_idx = self.vocab_start_idx + self.num_embeddings_per_partition
        self.weight = nn.Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim))
        self.weight.weight_loader = self.weight_loader

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(0)
        src = loaded_weight.narrow(0, self.tp_rank * shard, shard)
        assert param.data.size() == src.size()
        param.data.copy_(src)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.tp_size > 1:
            mask = (x >= self.vocab_start_idx) & (x < self.vocab_end_idx)
            x = mask * (x - self.vocab_start_idx)
        y = F.embedding(x, self.weight)
        if self.tp_size > 1:
            y = y * mask.unsqueeze(-1)
            tp.all_reduce(y)
        return y


class ParallelLMHead(VocabParallelEmbedding):

    def __init__(self, num_embeddings: int, embedding_dim: int, bias: bool = False):
        super().__init__(num_embeddings, embedding_dim)
        if bias:
            self.bias = nn.Parameter(torch.empty(self.num_embeddings_per_partition))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor):
        context = get_context()
        if context.is_prefill:
            last_indices = context.cu_seqlens_q[1:] - 1
            x = x[last_indices].contiguous()
        logits = F.linear(x, self.weight, self.bias)
        if self.tp_size > 1:
            parts = tp.gather_to_rank0(logits)
            logits = torch.cat(parts, -1) if self.tp_rank == 0 else None
        return logits

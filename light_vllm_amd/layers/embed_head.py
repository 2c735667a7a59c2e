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

    def greedy_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """argmax over the whole vocabulary without gathering it: every rank takes the maximum of
        its own shard, the ranks exchange one (value, index) pair per sequence (16 bytes instead of
        2 * vocab / tp), and the largest value wins, the lowest rank on equal values. That is
        exactly torch.argmax of the concatenated logits (first maximum), the token the reference's
        sampler returns for temperature 0 (layers/sampler.py:33, 64) from the logits its LM head
        gathers to rank 0 (layers/embed_head.py:175-195) - here every rank ends up with it.
        For steps whose sequences are all greedy; one rank: plain argmax."""
        context = get_context()
        if context.is_prefill:
            last_indices = context.cu_seqlens_q[1:] - 1
            x = x[last_indices].contiguous()
        logits = F.linear(x, self.weight, self.bias).float()
        value, index = logits.max(-1)               # first maximum of the shard
        if self.tp_size == 1:
            return index
        # bf16 logits and indices below 2^24 are exact in float64; one small tensor, one collective
        pair = torch.stack([value.double(), (index + self.vocab_start_idx).double()], -1)
        pairs = tp.all_gather(pair)                 # [tp, B, 2]
        best = pairs[..., 0].argmax(0)              # first maximum over ranks = lowest rank on ties
        return pairs[best, torch.arange(pairs.size(1), device=pairs.device), 1].long()

"""Qwen3 dense decoder, the immediate caller of the attention path.

Kept deliberately small: plain torch modules around `Attention` so the path can be exercised end to
end (greedy-token parity, tok/s). Structure, parameter names and rounding points follow the
reference model (lightvllm/models/qwen3.py:14-333): fused qkv projection split as [q | k | v] with
V left as a strided view (:94-95), per-head RMSNorm on q and k (:100-106), neox RoPE (:109), the
attention call (:113), row-parallel o_proj, pre-norm residual stream with fused add+RMSNorm.
"""
from __future__ import annotations

import torch
from torch import nn

from ..layers.activation import SiluAndMul
from ..layers.attention import Attention
from .. import ops
from ..layers.embed_head import ParallelLMHead, VocabParallelEmbedding
from ..layers.layernorm import RMSNorm
from ..layers.fused import use_fused
from ..layers.linear import MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear
from ..layers.rotary_embedding import get_rope
from ..layers import tp
from ..layers.tp import divide, tp_size
from ..utils.context import get_context


class Qwen3Attention(nn.Module):

    def __init__(self, hidden_size, num_heads, num_kv_heads, max_position=4096 * 32, head_dim=None,
                 rms_norm_eps=1e-6, qkv_bias=False, rope_theta=10000, rope_scaling=None):
        super().__init__()
        n = tp_size()
        self.total_num_heads = num_heads
        self.num_heads = divide(num_heads, n)
        self.total_num_kv_heads = num_kv_heads
        self.num_kv_heads = divide(num_kv_heads, n)
        self.head_dim = head_dim or hidden_size // num_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_proj = QKVParallelLinear(hidden_size, self.head_dim, num_heads, num_kv_heads, bias=qkv_bias)
        self.o_proj = RowParallelLinear(num_heads * self.head_dim, hidden_size, bias=False)
        self.rotary_emb = get_rope(self.head_dim, self.head_dim, max_position, rope_theta, rope_scaling)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads)
        self.q_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)
        self.k_norm = RMSNorm(self.head_dim, eps=rms_norm_eps)

    def forward(self, positions: torch.Tensor, hidden_states: torch.Tensor, reduce: bool = True) -> torch.Tensor:
        qkv = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        if use_fused(qkv):
            # q/k RMSNorm + RoPE in place on the qkv activation and the K/V cache scatter: one launch
            attn = self.attn
            ops.qk_norm_rope_store(q.view(-1, self.num_heads, self.head_dim),
                                   k.view(-1, self.num_kv_heads, self.head_dim),
                                   v.view(-1, self.num_kv_heads, self.head_dim), positions,
                                   self.q_norm.weight, self.k_norm.weight, self.q_norm.eps,
                                   self.rotary_emb.cos_sin_cache, attn.k_cache, attn.v_cache,
                                   get_context().slot_mapping)
            return self.o_proj(attn(q, k, v, kv_stored=True), reduce)
        q = self.q_norm(q.view(-1, self.num_heads, self.head_dim)).view(q.shape)
        k = self.k_norm(k.view(-1, self.num_kv_heads, self.head_dim)).view(k.shape)
        q, k = self.rotary_emb(positions, q, k)
        o = self.attn(q, k, v)
        return self.o_proj(o, reduce)


class Qwen3MLP(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str = "silu"):
        super().__init__()
        assert hidden_act == "silu"
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False)
        self.act_fn = SiluAndMul()

    def forward(self, x, reduce: bool = True):
        return self.down_proj(self.act_fn(self.gate_up_proj(x)), reduce)


class Qwen3DecoderLayer(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.self_attn = Qwen3Attention(
            hidden_size=config.hidden_size,
            num_heads=config.num_attention_heads,
            num_kv_heads=config.num_key_value_heads,
            max_position=config.max_position_embeddings,
            rms_norm_eps=config.rms_norm_eps,
            qkv_bias=getattr(config, "attention_bias", False),
            head_dim=getattr(config, "head_dim", None),
            rope_theta=getattr(config, "rope_theta", 1000000),
            rope_scaling=getattr(config, "rope_scaling", None),
        )
        self.mlp = Qwen3MLP(config.hidden_size, config.intermediate_size, config.hidden_act)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        """Same operations in the same order as the reference layer (models/qwen3.py:206-230). Under
        tensor parallelism the two row-parallel projections hand over their partial sums and the
        all-reduce runs together with the residual add + RMSNorm that consumes it
        (tp.all_reduce_add_rms_norm); the layer therefore RETURNS partial sums, which the next
        layer's input norm (or the model's final norm) reduces."""
        if residual is None:        # first layer: the embedding output is complete
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = tp.all_reduce_add_rms_norm(hidden_states, residual, self.input_layernorm)
        hidden_states = self.self_attn(positions, hidden_states, reduce=False)
        hidden_states, residual = tp.all_reduce_add_rms_norm(hidden_states, residual, self.post_attention_layernorm)
        hidden_states = self.mlp(hidden_states, reduce=False)
        return hidden_states, residual


class Qwen3Model(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen3DecoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, config.rms_norm_eps)

    def forward(self, input_ids, positions):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual)
        hidden_states, _ = tp.all_reduce_add_rms_norm(hidden_states, residual, self.norm)
        return hidden_states


class Qwen3ForCausalLM(nn.Module):
    packed_modules_mapping = {
        "q_proj": ("qkv_proj", "q"),
        "k_proj": ("qkv_proj", "k"),
        "v_proj": ("qkv_proj", "v"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config):
        super().__init__()
        self.model = Qwen3Model(config)
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def forward(self, input_ids, positions):
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states):
        return self.lm_head(hidden_states)

    def greedy_tokens(self, hidden_states):
        return self.lm_head.greedy_tokens(hidden_states)

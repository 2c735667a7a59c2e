"""Checkpoint loading: *.safetensors -> parameters through each parameter's `weight_loader`, with
the packed-module mapping (q/k/v -> qkv_proj, gate/up -> gate_up_proj) — the reference's
utils/loader.py:23-91 behaviour — plus a deterministic random initialiser for runs without weights.
"""
from __future__ import annotations

import os
import zlib
from glob import glob

import torch
from torch import nn


def default_weight_loader(param: nn.Parameter, loaded_weight: torch.Tensor):
    param.data.copy_(loaded_weight)


def load_model(model: nn.Module, path: str) -> None:
    from safetensors import safe_open
    packed = getattr(model, "packed_modules_mapping", {})
    for file in sorted(glob(os.path.join(path, "*.safetensors"))):
        with safe_open(file, "pt", "cpu") as f:
            for name in f.keys():
                for src, (dst, shard_id) in packed.items():
                    if src in name:
                        param = model.get_parameter(name.replace(src, dst))
                        param.weight_loader(param, f.get_tensor(name), shard_id)
                        break
                else:
                    try:
                        param = model.get_parameter(name)
                    except AttributeError:
                        continue   # e.g. lm_head.weight of a tied checkpoint
                    getattr(param, "weight_loader", default_weight_loader)(param, f.get_tensor(name))


def full_random_tensor(name: str, shape, seed: int, std: float) -> torch.Tensor:
    """The unsharded random value of checkpoint tensor `name`: a CPU fp32 normal draw seeded by
    (seed, crc32(name)), so every rank and every implementation derives identical weights."""
    g = torch.Generator(device="cpu").manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    return torch.randn(*shape, generator=g, dtype=torch.float32, device="cpu") * std


def random_checkpoint(hf_config, seed: int = 0, std: float = 0.02):
    """Yields (hf_name, tensor) pairs of a random Qwen3 checkpoint in HuggingFace naming
    (separate q/k/v and gate/up tensors; norm weights are ones plus a small perturbation)."""
    c = hf_config
    hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    H, KV, E, I, V = c.num_attention_heads, c.num_key_value_heads, c.hidden_size, c.intermediate_size, c.vocab_size

    def norm(name, n):
        return name, 1.0 + full_random_tensor(name, (n,), seed, 0.1)

    yield "model.embed_tokens.weight", full_random_tensor("model.embed_tokens.weight", (V, E), seed, std)
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        for nm, shape in ((p + "self_attn.q_proj.weight", (H * hd, E)), (p + "self_attn.k_proj.weight", (KV * hd, E)),
                          (p + "self_attn.v_proj.weight", (KV * hd, E)), (p + "self_attn.o_proj.weight", (E, H * hd)),
                          (p + "mlp.gate_proj.weight", (I, E)), (p + "mlp.up_proj.weight", (I, E)),
                          (p + "mlp.down_proj.weight", (E, I))):
            yield nm, full_random_tensor(nm, shape, seed, std)
        yield norm(p + "self_attn.q_norm.weight", hd)
        yield norm(p + "self_attn.k_norm.weight", hd)
        yield norm(p + "input_layernorm.weight", E)
        yield norm(p + "post_attention_layernorm.weight", E)
    yield norm("model.norm.weight", E)
    if not getattr(c, "tie_word_embeddings", False):
        yield "lm_head.weight", full_random_tensor("lm_head.weight", (V, E), seed, std)


def load_random(model: nn.Module, hf_config, seed: int = 0, std: float = 0.02) -> None:
    """Fill `model` (any TP shard of it) from random_checkpoint through the weight loaders."""
    packed = getattr(model, "packed_modules_mapping", {})
    for name, tensor in random_checkpoint(hf_config, seed, std):
        for src, (dst, shard_id) in packed.items():
            if src in name:
                param = model.get_parameter(name.replace(src, dst))
                param.weight_loader(param, tensor, shard_id)
                break
        else:
            param = model.get_parameter(name)
            getattr(param, "weight_loader", default_weight_loader)(param, tensor)


def load_random_on_device(model: nn.Module, seed: int = 0, std: float = 0.02) -> None:
    """Fast random fill on the parameters' own device (benchmarks only: values differ between
    devices and TP layouts, unlike load_random). Norm weights are set to one."""
    dev_gens = {}
    for name, param in model.named_parameters():
        if name.endswith("norm.weight") or "layernorm" in name:
            param.data.fill_(1.0)
            continue
        g = dev_gens.get(param.device)
        if g is None:
            g = dev_gens[param.device] = torch.Generator(device=param.device).manual_seed(seed)
        param.data.normal_(0.0, std, generator=g)

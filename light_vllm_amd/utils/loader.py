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


LAGCOPY_LAG = 257            # rotary-position lag the last layer's attention selects
LAGCOPY_FREQS = 8            # rotary pairs (the 8 fastest) carrying the position kernel
LAGCOPY_QK_GAIN = 4.0        # q_norm / k_norm weight of the last layer: peak score 11.3 * gain^2 = 181
LAGCOPY_COMMON = 128.0       # shared component of every embedding row (along one unit vector u)
LAGCOPY_CONTENT_STD = 4.0    # per-token part of an embedding row
LAGCOPY_OUT_GAIN = 40.0      # o_proj gain of the last layer: the copied row dominates the residual


def lagcopy_checkpoint(hf_config, seed: int = 0):
    """A seeded checkpoint whose greedy output is decided by attention WITH A MARGIN, for tests that
    demand token identity rather than tie-aware agreement. A plain random-init model cannot give
    that: its logits are Gaussian, the relative gap between the best two of 151936 is below 5 % on
    about every second step, and bf16 GEMM rounding (a few units in the last place of a logit)
    then picks the winner.

    All layers but the last are random as in random_checkpoint. The last layer is a lag-copy head:
      * every embedding row is COMMON * u + r_t (u a unit vector, r_t ~ N(0, CONTENT_STD^2));
      * q_proj = q0 u^T, k_proj = k0 u^T per head: after the per-head RMSNorm, q and k are the same
        two vectors for every token, living in the 8 fastest rotary pairs, q0 turned back by
        LAG * theta_f in pair f. After RoPE the score of (query at rotary position p, key at p') is
        181 * mean_f cos((p - p' - LAG) theta_f): 181 at the lag, at most 151 anywhere else, so
        the softmax puts weight 1 - 1e-11 on the one key whose rotary position is p - LAG;
      * v_proj = Q (I - u u^T) with Q orthogonal, o_proj = OUT_GAIN * Q^T regrouped per query head:
        the layer adds 40 x the selected token's normalised content to the residual, and the tied
        (or copied) LM head then scores that token ~4000 against < 700 for any other.
    So the next token is the token whose key sits LAG rotary positions back: prompt rows are at
    rotary position = index, decode rows at index + 1 (the reference's convention,
    lightvllm/engine/model_runner.py:405), which the copy pattern makes visible. Prompts must be
    at least LAG + 1 tokens long and generation at most LAG - 1 steps, so that the selected rotary
    position always exists (position len(prompt) itself never holds a key)."""
    c = hf_config
    hd = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    H, KV, E, V = c.num_attention_heads, c.num_key_value_heads, c.hidden_size, c.vocab_size
    assert hd == 128 and KV * hd >= E, "lag-copy layer needs head_dim 128 and kv width >= hidden size"
    last = f"model.layers.{c.num_hidden_layers - 1}."
    u = full_random_tensor("lagcopy.u", (E,), seed, 1.0)
    u = u / u.norm()
    # everything below is CPU fp32/fp64 whatever default device and dtype the caller has set
    cpu64 = dict(dtype=torch.float64, device="cpu")
    cpu32 = dict(dtype=torch.float32, device="cpu")
    theta = torch.tensor(float(c.rope_theta), **cpu64) ** (-torch.arange(LAGCOPY_FREQS, **cpu64) / (hd // 2))
    q0 = torch.zeros(hd, **cpu64)
    k0 = torch.zeros(hd, **cpu64)
    q0[:LAGCOPY_FREQS] = torch.cos(LAGCOPY_LAG * theta)          # pair f = (dim f, dim f + 64), neox layout
    q0[hd // 2:hd // 2 + LAGCOPY_FREQS] = -torch.sin(LAGCOPY_LAG * theta)
    k0[:LAGCOPY_FREQS] = 1.0
    Q, _ = torch.linalg.qr(full_random_tensor("lagcopy.Q", (KV * hd, E), seed, 1.0).double())   # orthonormal columns
    Q = Q.float()
    P = torch.eye(E, **cpu32) - torch.outer(u, u)
    G = H // KV
    for name, tensor in random_checkpoint(hf_config, seed):
        if name == "model.embed_tokens.weight" or name == "lm_head.weight":
            r = full_random_tensor("lagcopy.content", (V, E), seed, LAGCOPY_CONTENT_STD)
            yield name, r + LAGCOPY_COMMON * u
        elif name == last + "self_attn.q_proj.weight":
            yield name, torch.outer(q0.float(), u).repeat(H, 1)
        elif name == last + "self_attn.k_proj.weight":
            yield name, torch.outer(k0.float(), u).repeat(KV, 1)
        elif name == last + "self_attn.v_proj.weight":
            yield name, Q @ P
        elif name == last + "self_attn.o_proj.weight":
            blocks = [Q[(m // G) * hd:(m // G + 1) * hd].T * (LAGCOPY_OUT_GAIN / G) for m in range(H)]
            yield name, torch.cat(blocks, dim=1)
        elif name in (last + "self_attn.q_norm.weight", last + "self_attn.k_norm.weight"):
            yield name, torch.full((hd,), LAGCOPY_QK_GAIN, **cpu32)
        elif name == last + "input_layernorm.weight":
            yield name, torch.ones(E, **cpu32)
        else:
            yield name, tensor


def load_random(model: nn.Module, hf_config, seed: int = 0, std: float = 0.02, kind: str = "normal") -> None:
    """Fill `model` (any TP shard of it) from random_checkpoint (kind "normal") or lagcopy_checkpoint
    (kind "lagcopy") through the weight loaders."""
    packed = getattr(model, "packed_modules_mapping", {})
    source = lagcopy_checkpoint(hf_config, seed) if kind == "lagcopy" else random_checkpoint(hf_config, seed, std)
    for name, tensor in source:
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

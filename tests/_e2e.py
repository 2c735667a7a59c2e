"""Shared driver for the end-to-end greedy parity tests: steps this package's Scheduler +
ModelRunner through the reference run recorded in a golden file, teacher-forced, and compares
each step's argmax with the reference's token (tie-aware)."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import torch

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def qwen3_0p6b_config(layers=28, dtype=torch.bfloat16):
    return SimpleNamespace(
        vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=layers,
        num_attention_heads=16, num_key_value_heads=8, head_dim=128, hidden_act="silu",
        max_position_embeddings=40960, rms_norm_eps=1e-6, rope_theta=1000000, rope_scaling=None,
        attention_bias=False, tie_word_embeddings=True, torch_dtype=dtype)


def gqa4_config(layers=4, dtype=torch.bfloat16):
    """Qwen3-8B head geometry (32 q / 8 kv heads) on the 0.6B residual width."""
    cfg = qwen3_0p6b_config(layers, dtype)
    cfg.num_attention_heads = 32
    return cfg


def g8_config(layers=4, dtype=torch.bfloat16):
    """Per-rank head geometry of Qwen3-32B under TP = 8 (8 q / 1 kv head) on the 0.6B residual width."""
    cfg = qwen3_0p6b_config(layers, dtype)
    cfg.num_attention_heads = 8
    cfg.num_key_value_heads = 1
    return cfg


def golden_config(golden, dtype=torch.bfloat16):
    make = {"gqa4": gqa4_config, "g8": g8_config}.get(golden.get("shape"), qwen3_0p6b_config)
    return make(golden["layers"], dtype)


def golden_prompts(golden):
    return golden.get("prompts") or config1_prompts(golden["prompts_seed"], golden["num_prompts"], golden["prompt_len"])


def config1_prompts(seed=2026, n=4, length=256):
    g = torch.Generator().manual_seed(seed)
    return [torch.randint(0, 151936, (length,), generator=g).tolist() for _ in range(n)]


def bf16_ulp(x):
    """Spacing of bf16 values around x (8 significant bits)."""
    import math
    return 2.0 ** (math.floor(math.log2(max(abs(x), 1e-30))) - 7)


def is_tie(rec, j, tie_ulps):
    """The reference's own top-2 logits at this step lie within tie_ulps bf16 rounding steps of
    each other (each of the two can move by a step under a different summation order)."""
    return rec["gap"][j] <= tie_ulps * bf16_ulp(rec["top5_logits"][j][0]) + 1e-9


def lagcopy_expected(prompt, steps, lag=257):
    """What the lag-copy checkpoint (light_vllm_amd.utils.loader.lagcopy_checkpoint) must generate:
    the prefill's last row sits at rotary position len - 1 and copies the token `lag` positions
    back; decode rows sit at rotary position index + 1 (the reference feeds len(seq) as position),
    so generated token number i >= 1 copies prompt index len - lag + i."""
    n = len(prompt)
    return [prompt[n - 1 - lag]] + [prompt[n - lag + i] for i in range(1, steps)]


def replay(golden, runner, scheduler, tie_ulps, max_steps=None):
    """Returns (num_compared, tolerated, hard, max_dev): an argmax mismatch is tolerated only if
    the reference's own top-2 logits are tied (is_tie) and our token is in its top 5; max_dev is
    the largest |our logit - reference logit| over the reference's top-5 tokens of every step."""
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.sampling_params import SamplingParams
    from light_vllm_amd.utils.context import reset_context
    sp = SamplingParams(temperature=0.0, max_tokens=golden["max_tokens"], ignore_eos=True)
    prompts = golden_prompts(golden)
    seqs = [Sequence(p, sp) for p in prompts]
    index = {id(s): i for i, s in enumerate(seqs)}
    arrivals = golden.get("arrivals")       # {step: [sequence indices submitted before that step]}
    if not arrivals:
        for s in seqs:
            scheduler.add(s)
    compared, soft, hard, max_dev = 0, [], [], 0.0
    for step_no, rec in enumerate(golden["steps"][:max_steps]):
        for j in (arrivals or {}).get(str(step_no), []):
            scheduler.add(seqs[j])
        batch, is_prefill = scheduler.schedule()
        assert is_prefill == rec["prefill"], f"step {step_no}: phase differs"
        assert [index[id(s)] for s in batch] == rec["seqs"], f"step {step_no}: batch differs"
        if is_prefill and golden.get("cached_tokens_at_prefill"):   # prefix-cache hits as the reference's
            for s in batch:
                assert s.num_cached_tokens == golden["cached_tokens_at_prefill"][index[id(s)]]
        ids, pos = runner.prepare_prefill(batch) if is_prefill else runner.prepare_decode(batch)
        logits = runner.run_model(ids, pos, is_prefill).float()
        reset_context()
        ours = logits.argmax(-1).tolist()
        for j, (a, b) in enumerate(zip(ours, rec["tokens"])):
            compared += 1
            for tok, ref_logit in zip(rec["top5"][j], rec["top5_logits"][j]):
                max_dev = max(max_dev, abs(float(logits[j, tok]) - ref_logit))
            if a != b:
                item = (step_no, rec["seqs"][j], a, b, rec["gap"][j])
                if is_tie(rec, j, tie_ulps) and a in rec["top5"][j]:
                    soft.append(item)
                else:
                    hard.append(item)
        scheduler.postprocess(batch, rec["tokens"])     # teacher forcing with the reference tokens
    assert max_steps is not None or scheduler.is_finished()
    return compared, soft, hard, max_dev

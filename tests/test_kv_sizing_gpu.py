"""KV-cache sizing from free HBM, uncapped (num_kvcache_blocks = -1), on Qwen3-8B shapes: the
reference's formula (lightvllm/engine/model_runner.py:216-246) evaluated in 64-bit on a 288 GB
device gives a single allocation of more than 2^31 elements and 2^32 bytes. One prefill and one
decode step through the engine's own prepare_* / run_model with the sequences living in the
HIGHEST block ids must give the same logits, bit for bit, as the same sequences in the lowest."""
import json
import os
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
BLOCK_BYTES_8B = 37_748_736      # 2 (K,V) x 36 layers x 256 tokens x 8 kv heads x 128 x 2 B


def _qwen3_8b():
    return SimpleNamespace(vocab_size=151936, hidden_size=4096, intermediate_size=12288, num_hidden_layers=36,
                           num_attention_heads=32, num_key_value_heads=8, head_dim=128, hidden_act="silu",
                           max_position_embeddings=40960, rms_norm_eps=1e-6, rope_theta=1000000, rope_scaling=None,
                           attention_bias=False, tie_word_embeddings=False, torch_dtype=torch.bfloat16)


def test_uncapped_sizing_and_top_blocks(gpu):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.utils.context import reset_context
    torch.cuda.empty_cache()
    cfg = Config(None, hf_config=_qwen3_8b(), num_kvcache_blocks=-1, gpu_memory_utilization=0.9, max_num_seqs=16,
                 enforce_eager=True, random_seed=0, random_on_device=True)
    runner = ModelRunner(cfg, 0, [])
    z = runner.kv_sizing
    assert runner.kv_block_bytes() == z["block_bytes"] == BLOCK_BYTES_8B
    want = (int(z["total"] * 0.9) - z["used"] - z["peak"] + z["current"]) // BLOCK_BYTES_8B
    nblk = cfg.num_kvcache_blocks
    assert nblk == want == z["blocks_from_memory"] and z["cap"] == -1
    kv = runner.kv_cache
    assert tuple(kv.shape) == (2, 36, nblk, 256, 8, 128)
    assert kv.numel() > 2 ** 31 and kv.numel() * kv.element_size() > 2 ** 32
    free_after, total = torch.cuda.mem_get_info()
    line = dict(what="uncapped KV sizing, Qwen3-8B shapes, TP=1, gpu_memory_utilization 0.9", blocks=nblk,
                tokens=nblk * 256, kv_GiB=round(nblk * BLOCK_BYTES_8B / 2 ** 30, 1), total_GiB=round(total / 2 ** 30, 1),
                free_after_GiB=round(free_after / 2 ** 30, 1), **z)
    print("KV_SIZING " + json.dumps(line))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "kv_sizing.json"), "w") as f:
        json.dump(line, f)
    # a 288 GB device: a couple of hundred GiB of cache (any smaller card would still pass the formula check)
    if total > 250 * 2 ** 30:
        assert nblk * BLOCK_BYTES_8B > 200 * 2 ** 30

    g = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 151936, (n,), generator=g).tolist() for n in (300, 700)]

    def two_steps(tables):
        seqs = [Sequence(p) for p in prompts]
        for s, t in zip(seqs, tables):
            s.block_table = list(t)
        ids, pos = runner.prepare_prefill(seqs)
        logits_p = runner.run_model(ids, pos, True).float()
        reset_context()
        for s, tok in zip(seqs, logits_p.argmax(-1).tolist()):
            s.append_token(tok)
        ids, pos = runner.prepare_decode(seqs)
        logits_d = runner.run_model(ids, pos, False).float()
        reset_context()
        torch.cuda.synchronize()
        return logits_p.cpu(), logits_d.cpu()

    low = two_steps([[3, 0], [1, 4, 2]])
    top = two_steps([[nblk - 1, nblk - 4], [nblk - 2, nblk - 5, nblk - 3]])
    assert torch.equal(low[0], top[0]) and torch.equal(low[1], top[1])
    # the rows really went to the top of the allocation: last block of layer 35's V cache is not zero
    assert kv[1, 35, nblk - 1].abs().sum().item() > 0 and kv[0, 0, nblk - 5].abs().sum().item() > 0
    assert kv[0, 0, nblk // 2].abs().sum().item() == 0
    runner.exit()
    del runner, kv
    torch.cuda.empty_cache()

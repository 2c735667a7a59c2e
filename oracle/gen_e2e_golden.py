"""Writes tests/golden/e2e_config1.json: greedy decoding of BASELINE config 1 (Qwen3-0.6B shapes,
random-init, 4 x 256-token prompts, 64 greedy steps) by the REFERENCE's own model stack on CPU.

What runs from /root/reference: Qwen3ForCausalLM, the Attention.forward dispatcher, the TP layers,
Context, Scheduler, BlockManager, Sequence and Sampler. What is substituted (none of it can run
here: no flash_attn package, no nvcc, no GPU): the two flash_attn calls and store_kvcache are the
CPU oracle (oracle/attention_oracle.py), `lightvllm.kernels.cuda_kernels` is an empty module, and
ModelRunner.prepare_prefill/prepare_decode (model_runner.py:271-418) are the restatement in
oracle/host_oracle.py (no pin_memory/.cuda). Weights come from light_vllm_amd.utils.loader.random_checkpoint (seeded CPU
normal draws keyed by tensor name), pushed through the reference's own weight loaders.

Run in the build container only (TEST INFRASTRUCTURE; nothing in light_vllm_amd imports this):

    TORCHDYNAMO_DISABLE=1 python oracle/gen_e2e_golden.py [--layers N] [--steps K]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types
from types import SimpleNamespace

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from oracle import attention_oracle as orc  # noqa: E402
from oracle import host_oracle as host  # noqa: E402


def qwen3_0p6b_config(layers=28):
    return SimpleNamespace(
        vocab_size=151936, hidden_size=1024, intermediate_size=3072, num_hidden_layers=layers,
        num_attention_heads=16, num_key_value_heads=8, head_dim=128, hidden_act="silu",
        max_position_embeddings=40960, rms_norm_eps=1e-6, rope_theta=1000000, rope_scaling=None,
        attention_bias=False, tie_word_embeddings=True, torch_dtype=torch.bfloat16,
        kernel_backend="native")


def config1_prompts():
    g = torch.Generator().manual_seed(2026)
    return [torch.randint(0, 151936, (256,), generator=g).tolist() for _ in range(4)]


def gqa4_config(layers=4):
    """Qwen3-8B head geometry (32 query / 8 kv heads of 128: 4 query heads per kv head) on a narrow
    residual stream, so the reference stack still runs on CPU in seconds."""
    cfg = qwen3_0p6b_config(layers)
    cfg.num_attention_heads = 32
    return cfg


def prefix_prompts():
    """BASELINE config 5 in miniature: prompts that share whole 256-token blocks, submitted
    together, so BlockManager.allocate hits the prefix cache inside one prefill batch and the
    attention call takes its block-table (paged-prefix) prefill path. No length is a multiple
    of 256 (a fully cached prompt has zero query rows in the reference)."""
    g = torch.Generator().manual_seed(77)

    def rnd(n):
        return torch.randint(0, 151936, (n,), generator=g).tolist()

    base = rnd(512)
    return [base + rnd(88),            # 600 tokens: owns both shared blocks
            base[:256] + rnd(300),     # 556: hits the first block only
            base + rnd(40),            # 552: hits both
            rnd(300)]                  # no hit


def install_stubs():
    fa = types.ModuleType("flash_attn")

    def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k,
                               softmax_scale, causal, block_table=None):
        assert causal
        return orc.prefill_ref(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale, block_table).to(q.dtype)

    def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, causal):
        o = orc.decode_ref(q.squeeze(1), k_cache, v_cache, cache_seqlens, block_table, softmax_scale)
        return o.to(q.dtype).unsqueeze(1)

    fa.flash_attn_varlen_func = flash_attn_varlen_func
    fa.flash_attn_with_kvcache = flash_attn_with_kvcache
    sys.modules["flash_attn"] = fa
    ck = types.ModuleType("lightvllm.kernels.cuda_kernels")
    ck.cuda_swiglu = None
    sys.modules["lightvllm.kernels.cuda_kernels"] = ck


def long_prompts():
    """Two prompts beyond 2048 tokens (2100 and 2500): the prefill call takes the long-prompt
    kernel choice, and decode reads 9-10 blocks per sequence."""
    g = torch.Generator().manual_seed(78)
    return [torch.randint(0, 151936, (n,), generator=g).tolist() for n in (2100, 2500)]


def g8_config(layers=4):
    """Per-rank head geometry of Qwen3-32B under TP = 8 (8 query heads on 1 kv head of 128) on a
    narrow residual stream."""
    cfg = qwen3_0p6b_config(layers)
    cfg.num_attention_heads = 8
    cfg.num_key_value_heads = 1
    return cfg


def mixed_prompts():
    """BASELINE config 5 in miniature: six requests, four of them sharing a 1024-token prefix (whole
    or in part), arriving in two waves, so prefill steps (with prefix-cache hits on blocks that
    running sequences still hold) interleave with decode steps of the earlier requests.
    Returns (prompts, arrivals): arrivals[step] = indices submitted before that step is scheduled."""
    g = torch.Generator().manual_seed(80)

    def rnd(n):
        return torch.randint(0, 151936, (n,), generator=g).tolist()

    base = rnd(1024)
    prompts = [base + rnd(150), base + rnd(40), rnd(700),             # first wave
               base + rnd(300), rnd(300), base[:512] + rnd(100)]      # second wave, four decode steps later
    return prompts, {0: [0, 1, 2], 5: [3, 4, 5]}


def lagcopy_prompts():
    """Four prompts of 300 / 513 / 700 / 1024 random tokens for the lag-copy checkpoint
    (light_vllm_amd.utils.loader.lagcopy_checkpoint): all longer than its 257-position lag, of
    different block counts, one ending exactly on a block boundary."""
    g = torch.Generator().manual_seed(79)
    return [torch.randint(0, 151936, (n,), generator=g).tolist() for n in (300, 513, 700, 1024)]


def bf16_ulp(x):
    import math
    return 2.0 ** (math.floor(math.log2(max(abs(x), 1e-30))) - 7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "e2e_config1.json"))
    ap.add_argument("--shape", default="0.6b", choices=["0.6b", "gqa4", "g8"])
    ap.add_argument("--scenario", default="config1", choices=["config1", "prefix", "long", "lagcopy", "mixed"])
    ap.add_argument("--weights", default="normal", choices=["normal", "lagcopy"])
    ap.add_argument("--min-gap-ulps", type=float, default=0.0,
                    help="fail unless the top-2 logit gap of every recorded step is at least this many bf16 steps")
    args = ap.parse_args()

    install_stubs()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", world_size=1, rank=0)
    import lightvllm.layers.attention as ref_attn
    ref_attn.store_kvcache = orc.store_kvcache_ref
    from lightvllm.engine.scheduler import Scheduler
    from lightvllm.engine.sequence import Sequence
    from lightvllm.layers.sampler import Sampler
    from lightvllm.models.qwen3 import Qwen3ForCausalLM
    from lightvllm.sampling_params import SamplingParams
    from lightvllm.utils.context import reset_context, set_context
    from light_vllm_amd.utils.loader import load_random

    hf = {"0.6b": qwen3_0p6b_config, "gqa4": gqa4_config, "g8": g8_config}[args.shape](args.layers)
    torch.set_default_dtype(torch.bfloat16)
    model = Qwen3ForCausalLM(hf)
    torch.set_default_dtype(torch.float32)
    load_random(model, hf, args.seed, kind=args.weights)
    sampler = Sampler()

    block_size, num_blocks = 256, (24 if args.scenario in ("long", "lagcopy", "mixed") else 16)
    kv = torch.zeros(2, hf.num_hidden_layers, num_blocks, block_size, hf.num_key_value_heads, hf.head_dim,
                     dtype=torch.bfloat16)
    layer_id = 0
    for m in model.modules():
        if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
            m.k_cache, m.v_cache = kv[0, layer_id], kv[1, layer_id]
            layer_id += 1

    cfg = SimpleNamespace(max_num_seqs=512, max_num_batched_tokens=16384, eos=-1,
                          num_kvcache_blocks=num_blocks, kvcache_block_size=block_size)
    sch = Scheduler(cfg)
    sp = SamplingParams(temperature=0.0, max_tokens=args.steps, ignore_eos=True)
    arrivals = None
    if args.scenario == "mixed":
        prompts, arrivals = mixed_prompts()
    else:
        prompts = {"config1": config1_prompts, "prefix": prefix_prompts, "long": long_prompts,
                   "lagcopy": lagcopy_prompts}[args.scenario]()
    seqs = [Sequence(p, sp) for p in prompts]
    index = {id(s): i for i, s in enumerate(seqs)}
    if arrivals is None:
        for s in seqs:
            sch.add(s)

    # step inputs: the restated prepare_* of oracle/host_oracle.py, as CPU tensors
    def prepare_prefill(batch):
        r = host.prepare_prefill_ref(batch, block_size)
        bt = None if r["block_tables"] is None else torch.tensor(r["block_tables"], dtype=torch.int32)
        set_context(True, torch.tensor(r["cu_seqlens_q"], dtype=torch.int32),
                    torch.tensor(r["cu_seqlens_k"], dtype=torch.int32), r["max_seqlen_q"], r["max_seqlen_k"],
                    torch.tensor(r["slot_mapping"], dtype=torch.int32), None, bt)
        return torch.tensor(r["input_ids"], dtype=torch.int64), torch.tensor(r["positions"], dtype=torch.int64)

    def prepare_decode(batch):
        r = host.prepare_decode_ref(batch, block_size)
        set_context(False, slot_mapping=torch.tensor(r["slot_mapping"], dtype=torch.int32),
                    context_lens=torch.tensor(r["context_lens"], dtype=torch.int32),
                    block_tables=torch.tensor(r["block_tables"], dtype=torch.int32))
        return torch.tensor(r["input_ids"], dtype=torch.int64), torch.tensor(r["positions"], dtype=torch.int64)

    steps = []
    cached_at_prefill = [0] * len(seqs)
    t0 = time.time()
    with torch.inference_mode():
        while not sch.is_finished() or (arrivals and max(arrivals) >= len(steps)):
            for j in (arrivals or {}).get(len(steps), []):
                sch.add(seqs[j])
            batch, is_prefill = sch.schedule()
            if is_prefill:
                for s_ in batch:
                    cached_at_prefill[index[id(s_)]] = s_.num_cached_tokens
            ids, pos = prepare_prefill(batch) if is_prefill else prepare_decode(batch)
            logits = model.compute_logits(model(ids, pos)).float()
            temps = torch.tensor([s.temperature for s in batch], dtype=torch.float32)
            toks = sampler(logits.clone(), temps).tolist()
            reset_context()
            top_v, top_i = logits.topk(5, dim=-1)
            steps.append({"prefill": is_prefill, "seqs": [index[id(s)] for s in batch], "tokens": toks,
                          "gap": [round(float(x), 6) for x in (top_v[:, 0] - top_v[:, 1])],
                          "top5": top_i.tolist(),
                          "top5_logits": [[round(float(x), 5) for x in row] for row in top_v]})
            sch.postprocess(batch, toks)
            print(f"step {len(steps)} prefill={is_prefill} tokens={toks} t={time.time() - t0:.1f}s", flush=True)

    worst = min((rec["gap"][j] / bf16_ulp(rec["top5_logits"][j][0]), n, j) for n, rec in enumerate(steps)
                for j in range(len(rec["tokens"])))
    print(f"smallest top-2 gap: {worst[0]:.1f} bf16 steps of the top logit (step {worst[1]}, row {worst[2]})")
    if worst[0] < args.min_gap_ulps:
        raise SystemExit(f"gap {worst[0]:.1f} < required {args.min_gap_ulps} bf16 steps: not written")

    out = {"model": {"0.6b": "Qwen3-0.6B shapes, random init",
                     "gqa4": "Qwen3-8B head geometry (32q/8kv x 128) on hidden 1024, random init",
                     "g8": "Qwen3-32B TP=8 per-rank head geometry (8q/1kv x 128) on hidden 1024, random init"}[args.shape],
           "arrivals": None if arrivals is None else {str(k): v for k, v in arrivals.items()},
           "shape": args.shape, "scenario": args.scenario,
           "prompts": prompts if args.scenario != "config1" else None,
           "cached_tokens_at_prefill": [int(x) for x in cached_at_prefill],
           "layers": args.layers, "seed": args.seed,
           "weights": "light_vllm_amd.utils.loader.random_checkpoint(seed, std=0.02)" if args.weights == "normal"
                      else "light_vllm_amd.utils.loader.lagcopy_checkpoint(seed)",
           "weights_kind": args.weights, "min_gap_bf16_steps": round(worst[0], 2),
           "prompts_seed": 2026, "prompt_len": 256, "num_prompts": 4, "max_tokens": args.steps,
           "dtype": "bfloat16", "steps": steps,
           "completions": [list(s.completion_token_ids) for s in seqs]}
    with open(args.out, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote {args.out} ({os.path.getsize(args.out)} bytes) in {time.time() - t0:.1f}s")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

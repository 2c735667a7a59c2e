"""End-to-end throughput of this repository's engine on one MI355X: random-initialised Qwen3
shapes, synthetic prompts, greedy decoding with ignore_eos. Prints one JSON line per mode with the
engine's own counters (the reference prints the same two averages, llm_engine.py:292-295).

    python tools/e2e_bench.py --model 8b --batch 256 --prompt-len 1024 --new-tokens 64
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

SHAPES = {
    "0.6b": dict(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28, num_attention_heads=16,
                 num_key_value_heads=8, tie_word_embeddings=True),
    "8b": dict(hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
               num_key_value_heads=8, tie_word_embeddings=False),
    "32b": dict(hidden_size=5120, intermediate_size=25600, num_hidden_layers=64, num_attention_heads=64,
                num_key_value_heads=8, tie_word_embeddings=False),
}


def hf_config(name):
    return SimpleNamespace(vocab_size=151936, head_dim=128, hidden_act="silu", max_position_embeddings=40960,
                           rms_norm_eps=1e-6, rope_theta=1000000, rope_scaling=None, attention_bias=False,
                           torch_dtype=torch.bfloat16, **SHAPES[name])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="8b", choices=sorted(SHAPES))
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=1024)
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--modes", default="eager,graph")
    ap.add_argument("--blocks", type=int, default=0, help="cap on KV blocks (0 = size from free HBM)")
    ap.add_argument("--max-model-len", type=int, default=4096)
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel size: one process per GPU, RCCL over xGMI")
    ap.add_argument("--dist-port", type=int, default=2333, help="TCP port of the TP group's rendezvous")
    args = ap.parse_args()

    from light_vllm_amd.llm import LLM
    from light_vllm_amd.sampling_params import SamplingParams

    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 151936, (args.prompt_len,), generator=g).tolist() for _ in range(args.batch)]
    sp = SamplingParams(temperature=0.0, max_tokens=args.new_tokens, ignore_eos=True)
    need_blocks = args.batch * (-(-(args.prompt_len + args.new_tokens) // 256)) + 8
    for mode in args.modes.split(","):
        t0 = time.time()
        llm = LLM(None, hf_config=hf_config(args.model), enforce_eager=(mode == "eager"),
                  max_num_seqs=max(args.batch, 1), num_kvcache_blocks=args.blocks or max(need_blocks, 64),
                  max_model_len=args.max_model_len, max_num_batched_tokens=max(16384, args.max_model_len),
                  random_seed=0, random_on_device=True, tensor_parallel_size=args.tp,
                  dist_init_method=f"tcp://127.0.0.1:{args.dist_port}")
        t_init = time.time() - t0
        outs = llm.generate(prompts, sp)
        st = dict(llm.last_stats)
        blocks = llm.config.num_kvcache_blocks
        llm.exit()
        del llm
        torch.cuda.empty_cache()
        print(json.dumps({"model": f"Qwen3-{args.model} shapes (random init)", "mode": mode, "tp": args.tp,
                          "batch": args.batch, "prompt_len": args.prompt_len, "new_tokens": args.new_tokens,
                          "decode_tok_s": round(st["avg_decode_tok_s"], 1),
                          "prefill_tok_s": round(st["avg_prefill_tok_s"], 1),
                          "decode_ms_per_step": round(1e3 * st["decode_time"] / max(args.new_tokens - 1, 1), 3),
                          "kv_blocks": blocks, "init_s": round(t_init, 1),
                          "completions": len(outs)}), flush=True)


if __name__ == "__main__":
    main()

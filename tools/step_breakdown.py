"""Where a decode step's wall time goes (Qwen3-8B shapes, random weights, batch 256, hipGraph decode):
host preparation, GPU forward, sampling + device-to-host copy, scheduler bookkeeping."""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tools.e2e_bench import hf_config  # noqa: E402


def main():
    from light_vllm_amd.llm import LLM
    from light_vllm_amd.sampling_params import SamplingParams
    from light_vllm_amd.utils.context import reset_context
    B, L, steps = 256, 1024, 24
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 151936, (L,), generator=g).tolist() for _ in range(B)]
    sp = SamplingParams(temperature=0.0, max_tokens=steps + 8, ignore_eos=True)
    llm = LLM(None, hf_config=hf_config("8b"), enforce_eager=False, max_num_seqs=B,
              num_kvcache_blocks=B * 5 + 8, max_model_len=4096, max_num_batched_tokens=16384,
              random_seed=0, random_on_device=True)
    for p in prompts:
        llm.add_request(p, sp)
    while True:                      # prefill steps
        seqs, is_prefill = llm.scheduler.schedule()
        if not is_prefill:
            break
        llm.scheduler.postprocess(seqs, llm.model_runner.call("run", seqs, True))
    r = llm.model_runner
    acc = dict(schedule=0.0, prepare=0.0, forward=0.0, sample=0.0, postprocess=0.0)
    sync = torch.cuda.synchronize
    for i in range(steps):
        t0 = time.perf_counter()
        if i:
            seqs, is_prefill = llm.scheduler.schedule()
            assert not is_prefill
        t1 = time.perf_counter()
        ids, pos = r.prepare_decode(seqs)
        temps = r.prepare_sample(seqs)
        sync()
        t2 = time.perf_counter()
        logits = r.run_model(ids, pos, False)
        sync()
        t3 = time.perf_counter()
        toks = r.sampler(logits, temps, True).tolist()
        reset_context()
        t4 = time.perf_counter()
        llm.scheduler.postprocess(seqs, toks)
        t5 = time.perf_counter()
        if i >= 4:
            for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                acc[k] += d
    n = steps - 4
    print(json.dumps({k: round(1e3 * v / n, 3) for k, v in acc.items()} | {"unit": "ms per step", "batch": B}))
    llm.exit()


if __name__ == "__main__":
    main()

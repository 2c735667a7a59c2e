"""BASELINE config 1 on CPU ("plumbing, no GPU"): this package's host stack with the three
kernels substituted by the oracle, against the reference's stack run the same way
(oracle/gen_e2e_golden.py --layers 2 --steps 8 -> tests/golden/e2e_small.json)."""
import pytest
import torch

from tests._cpu_kernels import cpu_kernels
from tests._e2e import load_golden, qwen3_0p6b_config, replay


def _engine_parts(layers, seed):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    cfg = Config(None, hf_config=qwen3_0p6b_config(layers), device="cpu", num_kvcache_blocks=16,
                 max_model_len=512, max_num_batched_tokens=1024, random_seed=seed)
    runner = ModelRunner(cfg, 0, [])
    return cfg, runner, Scheduler(cfg)


def test_config1_small_matches_reference_stack():
    golden = load_golden("e2e_small.json")
    with cpu_kernels():
        cfg, runner, sch = _engine_parts(golden["layers"], golden["seed"])
        compared, soft, hard, _ = replay(golden, runner, sch, tie_ulps=2)
    assert compared == 4 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert len(soft) <= 1, f"too many tie flips: {soft}"


def test_generate_api_cpu():
    """LLM.generate end to end (free running) reproduces the reference completions."""
    from light_vllm_amd.llm import LLM
    from light_vllm_amd.sampling_params import SamplingParams
    from tests._e2e import config1_prompts
    golden = load_golden("e2e_small.json")
    with cpu_kernels():
        llm = LLM(None, hf_config=qwen3_0p6b_config(golden["layers"]), device="cpu", num_kvcache_blocks=16,
                  max_model_len=512, max_num_batched_tokens=1024, random_seed=golden["seed"])
        outs = llm.generate(config1_prompts(), SamplingParams(temperature=0.0, max_tokens=golden["max_tokens"],
                                                              ignore_eos=True))
        llm.exit()
    assert [o["token_ids"] for o in outs] == golden["completions"]
    assert llm.last_stats["decode_tokens"] == 4 * (golden["max_tokens"] - 1)


def test_prefix_cache_scenario_matches_reference_stack():
    """Prompts sharing whole blocks, submitted together (BASELINE config 5 in miniature, 32 q / 8
    kv heads): prefix-cache hits inside one prefill batch, paged-prefix prefill, then decode."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_prefix_gqa4.json")
    assert golden["cached_tokens_at_prefill"] == [0, 256, 512, 0]
    with cpu_kernels():
        cfg = Config(None, hf_config=golden_config(golden), device="cpu", num_kvcache_blocks=16,
                     max_model_len=1024, max_num_batched_tokens=4096, random_seed=golden["seed"])
        runner = ModelRunner(cfg, 0, [])
        compared, soft, hard, _ = replay(golden, runner, Scheduler(cfg), tie_ulps=2)
    assert compared == 4 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert len(soft) <= 2, f"too many tie flips: {soft}"


def test_long_prompt_scenario_matches_reference_stack():
    """Two prompts of 2100 and 2500 tokens in one prefill batch, then decode over 9-10 blocks per
    sequence (32 q / 8 kv heads), against the reference stack's recorded run."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_long_gqa4.json")
    with cpu_kernels():
        cfg = Config(None, hf_config=golden_config(golden), device="cpu", num_kvcache_blocks=24,
                     max_model_len=4096, max_num_batched_tokens=8192, random_seed=golden["seed"])
        runner = ModelRunner(cfg, 0, [])
        compared, soft, hard, _ = replay(golden, runner, Scheduler(cfg), tie_ulps=2)
    assert compared == 2 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert len(soft) <= 2, f"too many tie flips: {soft}"


def test_lagcopy_golden_is_the_copy_pattern_and_first_steps_replay():
    """The exact-identity fixture (tests/golden/e2e_lagcopy_256.json, recorded from the reference
    stack with utils.loader.lagcopy_checkpoint weights): its completions are the lag-copy pattern
    that the checkpoint's construction predicts, every recorded step has a top-2 gap of more than 8
    bf16 steps, and this package's CPU plumbing reproduces its first steps with no flip at all."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import bf16_ulp, golden_config, lagcopy_expected
    golden = load_golden("e2e_lagcopy_256.json")
    assert golden["weights_kind"] == "lagcopy" and golden["max_tokens"] == 256
    for prompt, completion in zip(golden["prompts"], golden["completions"]):
        assert completion == lagcopy_expected(prompt, 256)
    worst = min(rec["gap"][j] / bf16_ulp(rec["top5_logits"][j][0]) for rec in golden["steps"] for j in range(4))
    assert worst >= 8.0
    with cpu_kernels():
        cfg = Config(None, hf_config=golden_config(golden), device="cpu", num_kvcache_blocks=24, max_model_len=2048,
                     max_num_batched_tokens=4096, random_seed=golden["seed"], random_kind="lagcopy")
        runner = ModelRunner(cfg, 0, [])
        compared, soft, hard, _ = replay(golden, runner, Scheduler(cfg), tie_ulps=0, max_steps=3)
    assert compared == 12 and not soft and not hard


def test_mixed_prefill_decode_prefix_scenario_g8_matches_reference_stack():
    """BASELINE config 5 in miniature at its per-rank head geometry (Qwen3-32B under TP = 8: 8 q / 1
    kv head): six requests in two waves, four sharing a 1024-token prefix, so a prefill step with
    prefix-cache hits on blocks that running sequences still hold lands between decode steps."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_mixed_g8.json")
    assert golden["cached_tokens_at_prefill"] == [0, 1024, 0, 1024, 0, 512]
    assert [s["prefill"] for s in golden["steps"]].index(True, 1) == 5      # a prefill between decode steps
    with cpu_kernels():
        cfg = Config(None, hf_config=golden_config(golden), device="cpu", num_kvcache_blocks=24,
                     max_model_len=2048, max_num_batched_tokens=4096, random_seed=golden["seed"])
        runner = ModelRunner(cfg, 0, [])
        compared, soft, hard, _ = replay(golden, runner, Scheduler(cfg), tie_ulps=2)
    assert compared == 6 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert len(soft) <= 3, f"too many tie flips: {soft}"

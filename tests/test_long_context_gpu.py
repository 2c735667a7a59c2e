"""The engine at BASELINE config 4's context length: a 32 k-token prompt through ModelRunner
(block tables 125 entries wide, rotary positions up to 32 k), then a second prompt that shares its
first 124 blocks, so its prefill is 200 new tokens against 31 744 cached ones: the shape the
prefill kernel cuts into KV runs. The prefix-cached prefill and the decode steps after it must give
the logits of a run that never saw the first prompt (full prefill, nothing cached) to within bf16
noise, eager and with captured decode graphs. Reference behaviour being matched: a prefix hit only
changes where K/V are read from (lightvllm/layers/attention.py:155-165), never the result."""
import pytest
import torch

pytestmark = pytest.mark.gpu

BLOCK = 256
SHARED_BLOCKS = 124
TOL = 0.06      # max-abs logit difference between two bf16 runs that differ in summation order only


def _drive(prompt_groups, eager, heads):
    """One engine; the prompt groups are submitted one after another (a group runs to completion
    before the next is added). Returns per group the list of per-step logits (CPU fp32)."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.sampling_params import SamplingParams
    from light_vllm_amd.utils.context import reset_context
    from tests._e2e import g8_config, gqa4_config
    hf = gqa4_config(2) if heads == "32q/8kv" else g8_config(2)
    cfg = Config(None, hf_config=hf, num_kvcache_blocks=300, max_num_seqs=4, max_model_len=33024,
                 max_num_batched_tokens=33024, enforce_eager=eager, random_seed=5)
    from light_vllm_amd import _lib, ops
    runner, sch = ModelRunner(cfg, 0, []), Scheduler(cfg)
    # the runner reserved its split-KV workspace before it sized the cache: enough for the 200-row
    # prefill on 31 744 cached tokens (8 or 32 KV runs), and nothing allocates while serving
    allocations = ops.STATS["workspace_allocations"]
    hq = hf.num_attention_heads
    assert runner.workspace_bytes >= _lib.load().lv_prefill_workspace_bytes(
        200, hq, 128, _lib.load().lv_prefill_num_splits(1, 200, SHARED_BLOCKS * BLOCK + 200, hq))
    sp = SamplingParams(temperature=0.0, max_tokens=3, ignore_eos=True)
    out, cached = [], []
    for prompts, forced in prompt_groups:
        seqs = [Sequence(p, sp) for p in prompts]
        for s in seqs:
            sch.add(s)
        steps, step = [], 0
        while not sch.is_finished():
            batch, is_prefill = sch.schedule()
            if is_prefill:
                cached.append([s.num_cached_tokens for s in batch])
            ids, pos = runner.prepare_prefill(batch) if is_prefill else runner.prepare_decode(batch)
            logits = runner.run_model(ids, pos, is_prefill).float()
            reset_context()
            toks = logits.argmax(-1).tolist() if forced is None else forced[step]
            steps.append((logits.cpu(), toks))
            sch.postprocess(batch, toks)
            step += 1
        out.append(steps)
    assert ops.STATS["workspace_allocations"] == allocations, "the workspace grew while serving"
    runner.exit()
    return out, cached


@pytest.mark.parametrize("heads", ["32q/8kv", "8q/1kv"])
@pytest.mark.parametrize("eager", [True, False], ids=["eager", "hipgraph"])
def test_prefix_hit_at_32k_matches_uncached_run(gpu, eager, heads):
    g = torch.Generator().manual_seed(77)
    first = torch.randint(0, 151936, (32000,), generator=g).tolist()
    second = first[:SHARED_BLOCKS * BLOCK] + torch.randint(0, 151936, (200,), generator=g).tolist()
    # control: the second prompt alone, nothing cached
    (control,), cached_c = _drive([([second], None)], eager, heads)
    assert cached_c == [[0]]
    forced = [t for _, t in control]
    # the first prompt fills the cache, then the second one hits its first 124 blocks
    (_, hit), cached_h = _drive([([first], None), ([second], forced)], eager, heads)
    assert cached_h == [[0], [SHARED_BLOCKS * BLOCK]]
    assert len(hit) == len(control) == 3
    for (a, _), (b, _) in zip(control, hit):
        assert a.shape == b.shape == (1, 151936)
        assert torch.isfinite(b).all()
        dev = float((a - b).abs().max())
        assert dev <= TOL, f"logit deviation {dev}"


def test_prefix_hit_prefill_is_cut_into_runs(gpu):
    """The launch shape of the prefix-cached prefill above is one the library splits."""
    from light_vllm_amd import _lib
    lib = _lib.load()
    assert lib.lv_prefill_num_splits(1, 200, SHARED_BLOCKS * BLOCK + 200, 32) == 8
    assert lib.lv_prefill_num_splits(1, 200, SHARED_BLOCKS * BLOCK + 200, 8) == 32

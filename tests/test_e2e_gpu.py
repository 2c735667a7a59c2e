"""BASELINE config 1 end to end on the GPU: Qwen3-0.6B shapes, random init, 4 x 256-token prompts,
64 greedy steps through the HIP kernels, against the reference stack's recorded run
(tests/golden/e2e_config1.json, produced by oracle/gen_e2e_golden.py on CPU).

The reference run used CPU bf16 GEMMs and the fp32 oracle attention, this run uses hipBLASLt GEMMs
and the HIP kernels, so logits differ by bf16 rounding noise. The comparison is therefore
teacher-forced (each step continues from the reference's token) and tie-aware: our argmax must
equal the reference token unless the reference's own top-2 logits at that step are within
TIE_ULPS bf16 rounding steps of each other, in which case our token must still be one of its
top 5; and our logits at the reference's top-5 tokens must stay within MAX_LOGIT_DEV of its."""
import pytest
import torch

from tests._e2e import config1_prompts, is_tie, load_golden, qwen3_0p6b_config, replay

pytestmark = pytest.mark.gpu
TIE_ULPS = 2          # logits are bf16: ~3.0 at the top -> spacing 0.0156; their std is ~0.65
MAX_LOGIT_DEV = 0.1   # |ours - reference| on the reference's top-5 logits, any step


def _parts(golden, enforce_eager):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    cfg = Config(None, hf_config=qwen3_0p6b_config(golden["layers"]), num_kvcache_blocks=64,
                 max_num_seqs=16, enforce_eager=enforce_eager, random_seed=golden["seed"])
    runner = ModelRunner(cfg, 0, [])
    return cfg, runner, Scheduler(cfg)


@pytest.mark.parametrize("enforce_eager,fused", [(True, True), (False, True), (True, False)],
                         ids=["eager", "hipgraph", "eager-unfused"])
def test_config1_teacher_forced_parity(gpu, enforce_eager, fused):
    """fused=False routes q/k RMSNorm, RoPE, add+RMSNorm and SwiGLU through the op-by-op torch
    modules instead of the one-launch HIP kernels; both must meet the same bar."""
    from light_vllm_amd.layers import fused as fused_switch
    golden = load_golden("e2e_config1.json")
    old = fused_switch.set_enabled(fused)
    try:
        cfg, runner, sch = _parts(golden, enforce_eager)
        compared, soft, hard, max_dev = replay(golden, runner, sch, tie_ulps=TIE_ULPS)
    finally:
        fused_switch.set_enabled(old)
    print(f"compared {compared} tokens, max top-5 logit deviation {max_dev:.4f}, tie flips {soft}, hard mismatches {hard}")
    runner.exit()
    assert compared == 4 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert max_dev <= MAX_LOGIT_DEV, f"logit deviation {max_dev}"
    assert len(soft) <= compared // 20, f"too many tie flips: {soft}"


def test_config1_256_decode_steps(gpu):
    """The same four prompts carried through 256 decode steps (sequences grow from 256 to 512
    tokens and take a second KV block), hipGraph decode, teacher-forced and tie-aware as above."""
    golden = load_golden("e2e_config1_256.json")
    cfg, runner, sch = _parts(golden, enforce_eager=False)
    compared, soft, hard, max_dev = replay(golden, runner, sch, tie_ulps=TIE_ULPS)
    print(f"compared {compared} tokens, max top-5 logit deviation {max_dev:.4f}, {len(soft)} tie flips, hard mismatches {hard}")
    runner.exit()
    assert compared == 4 * 256
    assert not hard, f"token mismatches outside ties: {hard}"
    assert max_dev <= MAX_LOGIT_DEV, f"logit deviation {max_dev}"
    assert len(soft) <= compared // 20, f"too many tie flips: {soft}"


def test_generate_free_running(gpu):
    """LLM.generate (free running, hipGraph decode): every sequence must follow the reference
    completion at least up to the first step where the reference itself was tied (is_tie)."""
    from light_vllm_amd.llm import LLM
    from light_vllm_amd.sampling_params import SamplingParams
    golden = load_golden("e2e_config1.json")
    llm = LLM(None, hf_config=qwen3_0p6b_config(golden["layers"]), num_kvcache_blocks=64, max_num_seqs=16,
              enforce_eager=False, random_seed=golden["seed"])
    outs = llm.generate(config1_prompts(), SamplingParams(temperature=0.0, max_tokens=golden["max_tokens"],
                                                          ignore_eos=True))
    stats = llm.last_stats
    llm.exit()
    first_tie = [golden["max_tokens"]] * 4
    for step_no, rec in enumerate(golden["steps"]):
        for j, s in enumerate(rec["seqs"]):
            if is_tie(rec, j, TIE_ULPS):
                first_tie[s] = min(first_tie[s], step_no)
    for s in range(4):
        got, want = outs[s]["token_ids"], golden["completions"][s]
        agree = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), len(want))
        print(f"seq {s}: agrees for {agree}/{len(want)} tokens (first reference near-tie at step {first_tie[s]})")
        assert agree >= first_tie[s]
    assert stats["decode_tokens"] == 4 * (golden["max_tokens"] - 1)


@pytest.mark.parametrize("enforce_eager", [True, False], ids=["eager", "hipgraph"])
def test_prefix_cache_scenario_gqa4(gpu, enforce_eager):
    """BASELINE config 5 in miniature on Qwen3-8B head geometry (32 q / 8 kv heads: the matrix-core
    decode kernel): four prompts sharing whole 256-token blocks, submitted together, so the prefix
    cache hits inside one prefill batch (cached tokens [0, 256, 512, 0], as the reference's
    BlockManager decides) and the attention call takes the paged-prefix prefill path, then 23
    decode steps; teacher-forced and tie-aware against the reference stack's run."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_prefix_gqa4.json")
    cfg = Config(None, hf_config=golden_config(golden), num_kvcache_blocks=64, max_num_seqs=16,
                 enforce_eager=enforce_eager, random_seed=golden["seed"])
    runner = ModelRunner(cfg, 0, [])
    compared, soft, hard, max_dev = replay(golden, runner, Scheduler(cfg), tie_ulps=TIE_ULPS)
    print(f"compared {compared} tokens, max top-5 logit deviation {max_dev:.4f}, tie flips {soft}, hard mismatches {hard}")
    runner.exit()
    assert compared == 4 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert max_dev <= MAX_LOGIT_DEV, f"logit deviation {max_dev}"
    assert len(soft) <= compared // 10, f"too many tie flips: {soft}"


@pytest.mark.parametrize("enforce_eager", [True, False], ids=["eager", "hipgraph"])
def test_mixed_prefill_decode_prefix_scenario_g8(gpu, enforce_eager):
    """BASELINE config 5 in miniature at its per-rank head geometry (Qwen3-32B under TP = 8: 8 q / 1
    kv head per rank, the matrix-core decode kernel at group size 8): six requests in two waves,
    four of them sharing a 1024-token prefix. The second wave's prefill hits blocks that running
    sequences still hold (cached tokens [0, 1024, 0, 1024, 0, 512], as the reference's BlockManager
    decides), takes the paged-prefix prefill path between decode steps, and the batch then decodes
    with six sequences of different ages; teacher-forced and tie-aware against the reference
    stack's run."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_mixed_g8.json")
    cfg = Config(None, hf_config=golden_config(golden), num_kvcache_blocks=64, max_num_seqs=16, max_model_len=2048,
                 enforce_eager=enforce_eager, random_seed=golden["seed"])
    runner = ModelRunner(cfg, 0, [])
    compared, soft, hard, max_dev = replay(golden, runner, Scheduler(cfg), tie_ulps=TIE_ULPS)
    print(f"compared {compared} tokens, max top-5 logit deviation {max_dev:.4f}, tie flips {soft}, hard mismatches {hard}")
    runner.exit()
    assert compared == 6 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert max_dev <= MAX_LOGIT_DEV, f"logit deviation {max_dev}"
    assert len(soft) <= compared // 10, f"too many tie flips: {soft}"


@pytest.mark.parametrize("enforce_eager", [True, False], ids=["eager", "hipgraph"])
def test_long_prompt_scenario_gqa4(gpu, enforce_eager):
    """Prompts of 2100 and 2500 tokens in one prefill batch: the prefill call takes the kernel the
    library chooses for long prompts (the phase-locked 8-wave kernel), decode then reads 9-10
    blocks per sequence; teacher-forced and tie-aware against the reference stack's run."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from tests._e2e import golden_config
    golden = load_golden("e2e_long_gqa4.json")
    cfg = Config(None, hf_config=golden_config(golden), num_kvcache_blocks=64, max_num_seqs=16, max_model_len=4096,
                 enforce_eager=enforce_eager, random_seed=golden["seed"])
    runner = ModelRunner(cfg, 0, [])
    compared, soft, hard, max_dev = replay(golden, runner, Scheduler(cfg), tie_ulps=TIE_ULPS)
    print(f"compared {compared} tokens, max top-5 logit deviation {max_dev:.4f}, tie flips {soft}, hard mismatches {hard}")
    runner.exit()
    assert compared == 2 * golden["max_tokens"]
    assert not hard, f"token mismatches outside ties: {hard}"
    assert max_dev <= MAX_LOGIT_DEV, f"logit deviation {max_dev}"
    assert len(soft) <= 3, f"too many tie flips: {soft}"


def _stepwise_logits(prompts, params, eager, forced=None, max_num_seqs=16):
    """Drives scheduler + runner step by step; returns (per-step logits on CPU, per-step tokens,
    per-step (is_prefill, batch size)). With `forced`, continues from those tokens instead of its own."""
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.utils.context import reset_context
    from tests._e2e import gqa4_config
    cfg = Config(None, hf_config=gqa4_config(4), num_kvcache_blocks=64, max_num_seqs=max_num_seqs,
                 enforce_eager=eager, random_seed=3)
    runner, sch = ModelRunner(cfg, 0, []), Scheduler(cfg)
    for p, sp in zip(prompts, params):
        sch.add(Sequence(p, sp))
    logits_per_step, tokens_per_step, kinds = [], [], []
    step = 0
    while not sch.is_finished():
        batch, is_prefill = sch.schedule()
        ids, pos = runner.prepare_prefill(batch) if is_prefill else runner.prepare_decode(batch)
        logits = runner.run_model(ids, pos, is_prefill).float()
        reset_context()
        toks = logits.argmax(-1).tolist() if forced is None else forced[step]
        logits_per_step.append(logits.cpu())
        tokens_per_step.append(toks)
        kinds.append((is_prefill, len(batch)))
        sch.postprocess(batch, toks)
        step += 1
    runner.exit()
    return logits_per_step, tokens_per_step, kinds


def test_graph_padding_rows_match_eager(gpu):
    """Five sequences of different lengths decode in the 8-row graph bucket: three padding rows
    (slot -1, context length 0, their own index in the launch order) ride along. Logits of every
    step must match an eager runner fed the same tokens to within bf16 noise (the two differ in
    split count only)."""
    from light_vllm_amd.sampling_params import SamplingParams
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, 151936, (n,), generator=g).tolist() for n in (100, 257, 600, 31, 512)]
    params = [SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)] * 5
    eager, forced, _ = _stepwise_logits(prompts, params, True)
    graph, _, _ = _stepwise_logits(prompts, params, False, forced)
    assert len(eager) == len(graph) == 8
    for a, b in zip(eager, graph):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 0.06


def test_eager_overflow_batch_keeps_graph_workspace(gpu):
    """max_num_seqs 20 is not a multiple of 16: decode batches of 17-20 run eagerly above the
    largest graph bucket and need a larger split-KV workspace than any captured graph. Growing
    the workspace then must not free the buffer the graphs were captured with: after four
    sequences finish, the 16-row graph replays, and its logits must still match an all-eager run
    (a freed workspace shows up as garbage in the merged attention output)."""
    from light_vllm_amd import ops
    from light_vllm_amd.sampling_params import SamplingParams
    g = torch.Generator().manual_seed(12)
    prompts = [torch.randint(0, 151936, (int(n),), generator=g).tolist()
               for n in torch.randint(20, 700, (20,), generator=g)]
    params = [SamplingParams(temperature=0.0, max_tokens=3 if i % 5 == 0 else 7, ignore_eos=True) for i in range(20)]
    eager, forced, kinds = _stepwise_logits(prompts, params, True, max_num_seqs=20)
    ops._WORKSPACE.clear()                      # the graph run sizes the workspace on its own
    graph, _, kinds_g = _stepwise_logits(prompts, params, False, forced, max_num_seqs=20)
    assert kinds == kinds_g
    decode_sizes = [n for is_prefill, n in kinds if not is_prefill]
    assert 20 in decode_sizes and 16 in decode_sizes and decode_sizes.index(20) < decode_sizes.index(16)
    for a, b in zip(eager, graph):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 0.06
    # a later, larger direct call still must not drop a buffer that launches have used
    before = ops._WORKSPACE[gpu.index or 0]
    ops.reserve_decode_workspace(gpu, before.numel() + (1 << 20))
    assert any(t is before for t in ops._RETIRED)


def test_preemption_and_recompute_on_gpu(gpu):
    """Six sequences that outgrow a 14-block cache: the scheduler preempts (LIFO), the preempted
    sequences are prefilled again with their generated tokens, every block returns to the free
    list, and the completions start like those of a run with ample cache (greedy; where they
    part, at least 32 tokens in, is a near-tie of recomputed against incremental logits)."""
    from light_vllm_amd.llm import LLM
    from light_vllm_amd.sampling_params import SamplingParams
    from tests._e2e import gqa4_config
    g = torch.Generator().manual_seed(21)
    prompts = [torch.randint(0, 151936, (300,), generator=g).tolist() for _ in range(6)]
    sp = SamplingParams(temperature=0.0, max_tokens=300, ignore_eos=True)
    outs, preemptions = {}, {}
    for blocks in (64, 14):
        llm = LLM(None, hf_config=gqa4_config(4), num_kvcache_blocks=blocks, max_num_seqs=16, enforce_eager=False,
                  random_seed=3)
        count = [0]
        original = llm.scheduler.preempt

        def counted(seq, original=original, count=count):
            count[0] += 1
            return original(seq)
        llm.scheduler.preempt = counted
        outs[blocks] = [r["token_ids"] for r in llm.generate(prompts, sp)]
        preemptions[blocks] = count[0]
        assert len(llm.scheduler.block_manager.free_block_ids) == blocks
        llm.exit()
    assert preemptions[64] == 0 and preemptions[14] >= 1
    assert all(len(t) == 300 for t in outs[14])
    for a, b in zip(outs[64], outs[14]):
        agree = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), len(a))
        assert agree >= 32

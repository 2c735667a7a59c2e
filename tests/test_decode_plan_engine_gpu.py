"""The engine on a decode batch with one long context among short ones: ModelRunner.prepare_decode
builds a work list that cuts the long sequence into pieces (decode_plan.py), eager launches take it
directly, captured graphs take it through the second graph of the bucket. The logits must equal, to
bf16 summation-order noise, those of a runner that never plans (lightvllm/layers/attention.py:172-175
is one call either way), and the plan must really have been used."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 0.06


def _drive(prompts, eager, plan_on, steps=3):
    from light_vllm_amd.config import Config
    from light_vllm_amd.engine.model_runner import ModelRunner
    from light_vllm_amd.engine.scheduler import Scheduler
    from light_vllm_amd.engine.sequence import Sequence
    from light_vllm_amd.sampling_params import SamplingParams
    from light_vllm_amd.utils.context import get_context, reset_context
    from tests._e2e import gqa4_config
    cfg = Config(None, hf_config=gqa4_config(2), num_kvcache_blocks=80, max_num_seqs=8, max_model_len=8192,
                 max_num_batched_tokens=16384, enforce_eager=eager, random_seed=9)
    runner, sch = ModelRunner(cfg, 0, []), Scheduler(cfg)
    if not plan_on:
        runner._decode_plan = lambda lens: None
    sp = SamplingParams(temperature=0.0, max_tokens=steps, ignore_eos=True)
    for p in prompts:
        sch.add(Sequence(p, sp))
    out, planned = [], 0
    forced = None
    while not sch.is_finished():
        batch, is_prefill = sch.schedule()
        ids, pos = runner.prepare_prefill(batch) if is_prefill else runner.prepare_decode(batch)
        if not is_prefill and get_context().decode_plan is not None:
            planned += 1
            assert max(get_context().decode_plan.seq_splits if isinstance(get_context().decode_plan.seq_splits, list)
                       else get_context().decode_plan.seq_splits.tolist()) >= 4
        logits = runner.run_model(ids, pos, is_prefill).float()
        reset_context()
        out.append(logits.cpu())
        sch.postprocess(batch, logits.argmax(-1).tolist() if forced is None else forced[len(out) - 1])
    runner.exit()
    return out, planned


@pytest.mark.parametrize("eager", [True, False], ids=["eager", "hipgraph"])
def test_outlier_batch_planned_equals_unplanned(gpu, eager):
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, 151936, (n,), generator=g).tolist() for n in (6000, 300, 120, 700, 257)]
    want, n0 = _drive(prompts, eager, plan_on=False)
    got, n1 = _drive(prompts, eager, plan_on=True)
    assert n0 == 0 and n1 >= 2, "the decode steps of this batch must run from a work list"
    assert len(want) == len(got)
    for a, b in zip(want, got):
        assert a.shape == b.shape and torch.isfinite(b).all()
        assert float((a - b).abs().max()) <= TOL
        assert torch.equal(a.argmax(-1), b.argmax(-1)) or float((a - b).abs().max()) <= TOL

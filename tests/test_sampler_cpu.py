"""Sampler (reference: lightvllm/layers/sampler.py:16-64): argmax for temperature 0, exponential-race
sampling otherwise, row by row; and the host-known all-greedy shortcut returns the same tokens."""
import torch

from light_vllm_amd.layers.sampler import Sampler


def test_greedy_rows_and_shortcut_agree():
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(7, 1000, generator=g).to(torch.bfloat16)
    zero = torch.zeros(7)
    s = Sampler()
    want = logits.float().argmax(-1)
    assert torch.equal(s(logits, zero), want)
    assert torch.equal(s(logits, zero, True), want)


def test_mixed_rows_keep_greedy_rows_exact():
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(6, 500, generator=g)
    temps = torch.tensor([0.0, 1.0, 0.0, 0.7, 0.0, 1.3])
    torch.manual_seed(5)
    out = Sampler()(logits, temps)
    want = logits.argmax(-1)
    for r in (0, 2, 4):
        assert out[r] == want[r]
    assert bool(((out >= 0) & (out < 500)).all())


def test_sampled_rows_follow_the_softmax():
    """20000 draws from a 5-token distribution at temperature 1 and 0.5: frequencies within 4 sigma."""
    probs = torch.tensor([0.5, 0.25, 0.15, 0.07, 0.03])
    n = 20000
    for temp in (1.0, 0.5):
        logits = (probs.log() * 1.0).repeat(n, 1)
        torch.manual_seed(9)
        out = Sampler()(logits, torch.full((n,), temp))
        target = torch.softmax(probs.log() / temp, -1)
        freq = torch.bincount(out, minlength=5).float() / n
        sigma = (target * (1 - target) / n).sqrt()
        assert bool(((freq - target).abs() <= 4 * sigma + 1e-4).all()), (temp, freq, target)

"""Greedy for temperature 0, exponential-race (Gumbel-max) sampling otherwise
(reference: lightvllm/layers/sampler.py:16-64)."""
import torch
from torch import nn


class Sampler(nn.Module):

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor, all_greedy: bool = False) -> torch.Tensor:
        """all_greedy: the caller knows on the host that every temperature is 0; the result is then
        the argmax alone (what the select below returns for such rows), without the softmax, noise
        and second argmax over the [batch, vocab] logits."""
        logits = logits.to(torch.float)
        greedy = logits.argmax(-1)
        if all_greedy:
            return greedy
        logits = logits / temperatures.unsqueeze(1)
        probs = torch.softmax(logits, -1, dtype=torch.float)
        noise = torch.empty_like(probs).exponential_(1) + 1e-10
        sampled = (probs / noise).argmax(-1)
        return torch.where(temperatures == 0, greedy, sampled)

"""Host-side work list of a paged-decode attention call on a batch of unequal lengths (the
`work_items` / `seq_splits` arguments of lv_decode_attn_planned, include/lv_attention.h).

The caller of the reference's flash_attn_with_kvcache (lightvllm/layers/attention.py:172-175) has
every context length on the host when it builds context_lens (engine/model_runner.py:383-418);
the kernels of lv_decode_attn see them only on the device and must give every sequence the same
split count. From the lengths this module cuts each sequence into pieces of about equal size,
longest pieces first, so that no workgroup streams far longer than the others and the launch does
not end on a few long stragglers.

Pure Python on plain ints; the result of the attention call never depends on the plan beyond the
fp32 summation order of the split merge.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

SPLIT_UNIT = 64          # split boundaries of the decode kernels are multiples of this many tokens
GROUP = 32               # items x all kv heads per launch group (the kernels' grouped order)
MIN_PIECE = 1024         # never cut below this many tokens per piece: a workgroup's fixed costs (its first
                         # loads, the merge of its four waves) weigh as much as ~150 tokens of streaming,
                         # and 256 sequences of 1..4096 tokens ran 20 % slower cut into 256-token pieces
COMPUTE_UNITS = 256


@dataclass
class DecodePlan:
    """items: flat int32 list {used, group, (sequence, split | splits << 16) ...} padded to `capacity`
    items; seq_splits: per-sequence split count; max_splits: split capacity a workspace row needs."""
    items: object
    seq_splits: object
    capacity: int
    max_splits: int
    used: int
    piece_tokens: int

    def to(self, device):
        import torch
        it = torch.tensor(self.items, dtype=torch.int32)
        ss = torch.tensor(self.seq_splits, dtype=torch.int32)
        if torch.device(device).type != "cpu":
            it = it.pin_memory().to(device, non_blocking=True)
            ss = ss.pin_memory().to(device, non_blocking=True)
        return DecodePlan(it, ss, self.capacity, self.max_splits, self.used, self.piece_tokens)


def piece_size(context_lens: Sequence[int], num_kv_heads: int, group_heads: int, pieces_per_slot: float = 5.0) -> int:
    """Tokens per piece: large enough that a compute-unit slot sees about pieces_per_slot pieces over
    the launch (the launch then ends within a fraction of its length after its last piece starts),
    never below MIN_PIECE, a multiple of the kernels' 64-token split unit."""
    slots = COMPUTE_UNITS * (3 if group_heads <= 2 else 2)
    want = int(sum(context_lens) * num_kv_heads / (slots * pieces_per_slot))
    return max(MIN_PIECE, -(-want // SPLIT_UNIT) * SPLIT_UNIT)


def worth_planning(context_lens: Sequence[int], num_kv_heads: int, group_heads: int) -> bool:
    """A plan pays when it cuts something: when the longest sequence is more than one and a half
    pieces long while the batch is too large for lv_decode_attn's one split count to cut anyone
    (that count is sized for the launch shape, not for the lengths). Measured, 256 sequences of
    1..1024 tokens with four of 16384 among them, 32q/8kv: 332 us in one pass, 131 us planned; with
    one of 32768: 573 -> 113 us. Batches without such outliers gain nothing from a list."""
    if len(context_lens) < 2:
        return False
    piece = piece_size(context_lens, num_kv_heads, group_heads)
    hi = max(context_lens)
    return hi > piece + piece // 2 and hi > 2 * (sum(context_lens) // len(context_lens))


def build_plan(context_lens: Sequence[int], num_kv_heads: int, group_heads: int,
               max_splits: int = 64, capacity: int = 0, pieces_per_slot: float = 5.0,
               piece_tokens: int = 0) -> DecodePlan:
    """context_lens: python ints, one per sequence (0 allowed: a padding row, served by one item
    that writes its zero row). group_heads = query heads per kv head (decides which kernel runs and
    how many workgroups a compute unit holds). max_splits: most pieces per sequence (the workspace
    row capacity the caller has room for). capacity: items the list may hold at most (0: as many as
    the plan needs; a captured graph passes its fixed capacity and the pieces grow until they fit).
    piece_tokens: force the piece size (0: chosen so that a compute-unit slot sees about
    pieces_per_slot pieces over the launch, which bounds the tail of the launch to a fraction of it).
    """
    lens = [int(x) for x in context_lens]
    B = len(lens)
    assert B > 0 and max_splits >= 1
    if piece_tokens <= 0:
        piece = piece_size(lens, num_kv_heads, group_heads, pieces_per_slot)
    else:
        piece = max(SPLIT_UNIT, -(-piece_tokens // SPLIT_UNIT) * SPLIT_UNIT)

    def cut(piece):
        # nearest count: pieces of two thirds to four thirds of `piece` tokens
        return [min(max_splits, max(1, (n + piece // 2) // piece)) for n in lens]

    splits = cut(piece)
    if capacity > 0:
        assert capacity >= B, "the list must hold one item per sequence at least"
        while sum(splits) > capacity:
            piece *= 2
            splits = cut(piece)
    # the kernels cut a sequence of n tokens into n_b runs of whole 64-token units, as even as the
    # units allow: piece sizes for the ordering below
    entries = []
    for b, (n, nb) in enumerate(zip(lens, splits)):
        units = -(-n // SPLIT_UNIT)
        for s in range(nb):
            lo, hi = s * units // nb, (s + 1) * units // nb
            tokens = max(0, min(hi * SPLIT_UNIT, n) - lo * SPLIT_UNIT)
            entries.append((tokens, b, s, nb))
    entries.sort(key=lambda e: -e[0])            # stable: equal pieces keep sequence order
    used = len(entries)
    cap = capacity if capacity > 0 else used
    items: List[int] = [used, GROUP]
    for _, b, s, nb in entries:
        items += [b, s | (nb << 16)]
    items += [-1, 0] * (cap - used)
    return DecodePlan(items, splits, cap, max(splits), used, piece)


def natural_plan(batch: int, num_splits: int, capacity: int = 0) -> DecodePlan:
    """The list that reproduces lv_decode_attn's own mapping for `batch` sequences cut into
    num_splits each (split fastest, then sequence, one group): for callers whose launch is a captured
    graph with a work list and whose batch of the step has nearly equal lengths."""
    used = batch * num_splits
    cap = capacity if capacity > 0 else used
    assert cap >= used
    items: List[int] = [used, max(used, 1)]
    for b in range(batch):
        for s in range(num_splits):
            items += [b, s | (num_splits << 16)]
    items += [-1, 0] * (cap - used)
    return DecodePlan(items, [num_splits] * batch, cap, num_splits, used, 0)

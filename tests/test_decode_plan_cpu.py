"""Host-built work lists of the paged-decode call (light_vllm_amd/decode_plan.py): every split of
every sequence exactly once, counts consistent with seq_splits, longest pieces first, capacity
respected. Pure Python; the GPU side is tests/test_decode_plan_gpu.py."""
import random

import pytest

from light_vllm_amd.decode_plan import MIN_PIECE, SPLIT_UNIT, build_plan, natural_plan, piece_size, worth_planning


def _decode_items(plan):
    used, group = plan.items[0], plan.items[1]
    body = plan.items[2:]
    assert len(body) == 2 * plan.capacity
    live = [(body[2 * i], body[2 * i + 1] & 0xffff, body[2 * i + 1] >> 16) for i in range(used)]
    dead = [(body[2 * i], body[2 * i + 1]) for i in range(used, plan.capacity)]
    return used, group, live, dead


def _piece_tokens(n, s, nb):
    units = -(-n // SPLIT_UNIT)
    lo, hi = s * units // nb, (s + 1) * units // nb
    return max(0, min(hi * SPLIT_UNIT, n) - lo * SPLIT_UNIT)


@pytest.mark.parametrize("seed", range(8))
def test_plan_covers_every_split_once(seed):
    rng = random.Random(seed)
    B = rng.choice([1, 3, 40, 256])
    lens = [rng.choice([0, 1, 63, 64, 65, 255, 256, 257, rng.randint(1, 40000)]) for _ in range(B)]
    hkv, g = rng.choice([(8, 2), (8, 4), (1, 8), (2, 4)])
    cap = rng.choice([0, B + 64, B + 512])
    plan = build_plan(lens, hkv, g, max_splits=rng.choice([1, 4, 16, 64]), capacity=cap)
    used, group, live, dead = _decode_items(plan)
    assert used == len(live) == sum(plan.seq_splits) and group >= 1
    assert plan.capacity == (cap or used) and used <= plan.capacity
    assert all(b == -1 for b, _ in dead)
    seen = set()
    for b, s, nb in live:
        assert 0 <= b < B and nb == plan.seq_splits[b] and 0 <= s < nb <= plan.max_splits
        assert (b, s) not in seen
        seen.add((b, s))
    assert len(seen) == sum(plan.seq_splits)            # every (sequence, split) is there
    sizes = [_piece_tokens(lens[b], s, nb) for b, s, nb in live]
    assert sizes == sorted(sizes, reverse=True)         # longest pieces start first
    for b, n in enumerate(lens):                        # no piece far below the floor
        if plan.seq_splits[b] > 1:
            assert n / plan.seq_splits[b] > MIN_PIECE / 2


def test_plan_cuts_long_sequences_only():
    lens = [32768] + [700] * 63
    plan = build_plan(lens, 8, 4, max_splits=64)
    assert plan.seq_splits[0] >= 16 and all(x == 1 for x in plan.seq_splits[1:])
    # the same batch under a tight list: pieces grow until the list fits
    tight = build_plan(lens, 8, 4, max_splits=64, capacity=70)
    assert sum(tight.seq_splits) <= 70 and tight.seq_splits[0] >= 2


def test_natural_plan_is_the_kernels_own_mapping():
    plan = natural_plan(5, 3, capacity=20)
    used, group, live, dead = _decode_items(plan)
    assert used == 15 and group == 15 and len(dead) == 5
    assert live == [(b, s, 3) for b in range(5) for s in range(3)]


def test_worth_planning():
    assert not worth_planning([1024] * 256, 8, 4)
    assert not worth_planning([1000, 900, 800], 8, 4)
    assert not worth_planning([1024, 10], 8, 4)               # nothing would be cut
    assert not worth_planning([32768] * 32, 8, 4)             # equal lengths: one split count serves
    assert worth_planning([32768] + [700] * 255, 8, 4)
    assert worth_planning([16384] * 4 + [500] * 252, 8, 2)
    assert not worth_planning([], 8, 4) and not worth_planning([50000], 8, 4)
    assert piece_size([1000] * 10, 8, 4) == MIN_PIECE and piece_size([32768] * 32, 8, 4) % SPLIT_UNIT == 0

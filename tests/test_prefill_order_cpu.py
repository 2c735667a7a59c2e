"""The host-built launch order of prefill query tiles (light_vllm_amd/prefill_order.py): every tile
once, columns balanced, a sequence confined to few columns."""
import random

import pytest

from light_vllm_amd.prefill_order import build_tile_order, order_columns, tile_weight


def _decode(order, max_q):
    qtiles = -(-max_q // order.tile_rows)
    cols = [[] for _ in range(order.cols)]
    for i, code in enumerate(order.table):
        if code >= 0:
            cols[i % order.cols].append((code // qtiles, code % qtiles))
    return cols


@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (8, 2), (16, 4), (32, 8), (9, 3)])
@pytest.mark.parametrize("rows", [128, 256])
def test_every_tile_exactly_once(Hq, Hkv, rows):
    rng = random.Random(5)
    for _ in range(30):
        n = rng.randint(1, 12)
        lens_k = [rng.choice([0, 1, 127, 128, 129, 255, 256, 257, rng.randint(1, 9000)]) for _ in range(n)]
        lens_q = [rng.randint(0, lk) for lk in lens_k]
        max_q = max(lens_q)
        order = build_tile_order(lens_q, lens_k, max_q, rows, Hkv)
        want = sorted((b, t) for b, lq in enumerate(lens_q) for t in range(-(-lq // rows)))
        if not want:
            assert order is None
            continue
        assert order.cols == order_columns(Hkv) and len(order.table) % order.cols == 0
        got = sorted(x for col in _decode(order, max_q) for x in col)
        assert got == want


def test_tp8_shard_of_four_equal_prompts():
    """4 x 4096 tokens, 8q/1kv: eight columns, each half of one sequence at equal weight."""
    order = build_tile_order([4096] * 4, [4096] * 4, 4096, 256, 1)
    cols = _decode(order, 4096)
    assert len(cols) == 8 and all(len(c) == 8 for c in cols)
    for c in cols:
        assert len({b for b, _ in c}) == 1
        assert sum(tile_weight(4096, 4096, t, 256) for _, t in c) == 272
    first = [t for _, t in cols[0]]
    assert first == sorted(first, reverse=True)      # heaviest (last rows) first


def test_columns_are_balanced_on_ragged_lengths():
    lens = [4096, 3000, 1024, 8264]
    order = build_tile_order(lens, lens, max(lens), 256, 1)
    loads = [sum(tile_weight(lens[b], lens[b], t, 256) for b, t in col) for col in _decode(order, max(lens))]
    assert max(loads) - min(loads) <= 2 * max(tile_weight(8264, 8264, 32, 256), 1)
    for col in _decode(order, max(lens)):
        w = [tile_weight(lens[b], lens[b], t, 256) for b, t in col]
        assert w == sorted(w, reverse=True)


def test_bottom_right_alignment_of_weights():
    # 100 new rows on 4096 cached: every tile walks the whole prefix
    assert tile_weight(100, 4196, 0, 128) == (4196 - 1) // 64 + 1
    # packed: tile t of 128 rows sees keys up to its last row
    assert tile_weight(1000, 1000, 0, 128) == 2 and tile_weight(1000, 1000, 7, 128) == 16


def test_item_table_covers_every_tile_once_and_balances_lanes():
    """Item tables of the persistent long-prompt kernel: every query tile of every sequence appears in
    exactly one list with the values cu_seqlens imply, lists are contiguous from depth 0, heaviest
    first, and carry near-equal numbers of key tiles."""
    from light_vllm_amd.prefill_order import build_item_table, order_columns, order_lanes, tile_weight
    for lens_q, lens_k, Hq, Hkv, paged in (([4096] * 4, [4096] * 4, 32, 8, False),
                                           ([4096, 3000, 1024, 8264], [4096, 3000, 1024, 8264], 32, 8, False),
                                           ([1976, 256, 700, 2088], [3000, 2304, 700, 2600], 8, 1, True),
                                           ([300, 1], [300, 1], 4, 1, False)):
        order = build_item_table(lens_q, lens_k, max(lens_q), 256, Hq, Hkv, paged)
        cols, lanes = order_columns(Hkv), order_lanes(Hq, Hkv)
        assert order.cols == cols and order.lanes == lanes and order.paged == paged
        assert lanes == max(1, 256 // (Hq * cols))
        tab = order.table
        assert len(tab) % (8 * cols * lanes) == 0
        depth = len(tab) // (8 * cols * lanes)
        cu_q = [0]
        cu_k = [0]
        for a, b in zip(lens_q, lens_k):
            cu_q.append(cu_q[-1] + a)
            cu_k.append(cu_k[-1] + b)
        seen, loads = set(), []
        for c in range(cols):
            for r in range(lanes):
                ended, load, last_w = False, 0, None
                for k in range(depth):
                    e = tab[((k * lanes + r) * cols + c) * 8:][:8]
                    if e[1] <= 0:
                        ended = True
                        continue
                    assert not ended, "a list must be contiguous from depth 0"
                    qrow0, nrows, lim0, kref, lk = e[:5]
                    b = next(i for i in range(len(lens_q)) if cu_q[i] <= qrow0 < cu_q[i + 1])
                    q0 = qrow0 - cu_q[b]
                    assert q0 % 256 == 0 and nrows == min(256, lens_q[b] - q0)
                    assert lim0 == q0 + lens_k[b] - lens_q[b] and lk == lens_k[b]
                    assert kref == (b if paged else cu_k[b])
                    assert (b, q0) not in seen
                    seen.add((b, q0))
                    w = tile_weight(lens_q[b], lens_k[b], q0 // 256, 256)
                    assert last_w is None or w <= last_w, "heaviest first inside a list"
                    last_w = w
                    load += w
                loads.append(load)
        assert len(seen) == sum(-(-n // 256) for n in lens_q)
        if cols == 1 and len(seen) >= 4 * lanes:
            assert max(loads) - min(loads) <= 64 + max(loads) // 10      # longest-processing-time-first keeps the lists close

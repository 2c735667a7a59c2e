"""Randomised shape sweep: the HIP decode and prefill kernels against the CPU fp32 oracle on
seeded random combinations of batch size, lengths (short, block-boundary and long), head
grouping, block size, split count, launch order, host-built work lists and tile tables, and kernel choice — the corners between the
hand-picked cases of the other GPU tests. Tolerance as there: max-abs <= 2e-2 on bf16 outputs."""
import os

import pytest
import torch

from oracle.attention_oracle import decode_ref, prefill_ref
from tests._cases import make_decode_case, make_prefill_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-2
GROUPS = [(1, 1), (2, 1), (4, 1), (8, 1), (2, 2), (8, 4), (16, 8), (32, 8), (16, 4), (64, 8),
          (3, 1), (5, 1), (40, 8), (6, 1), (7, 1), (12, 2)]


def _lengths(g, n, hi):
    """Mix of tiny, block-boundary and arbitrary lengths up to hi."""
    special = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513]
    out = []
    for _ in range(n):
        if torch.rand((), generator=g).item() < 0.4:
            out.append(min(special[int(torch.randint(0, len(special), (), generator=g))], hi))
        else:
            out.append(int(torch.randint(1, hi + 1, (), generator=g)))
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("LV_FUZZ_DECODE", "40"))))   # more seeds for a soak run
def test_fuzz_decode(gpu, monkeypatch, seed):
    from light_vllm_amd import ops
    g = torch.Generator().manual_seed(1000 + seed)
    Hq, Hkv = GROUPS[int(torch.randint(0, len(GROUPS), (), generator=g))]
    block = [32, 64, 128, 256][int(torch.randint(0, 4, (), generator=g))]
    B = int(torch.randint(1, 41, (), generator=g))
    lens = _lengths(g, B, 700)
    if torch.rand((), generator=g).item() < 0.3:
        lens[int(torch.randint(0, B, (), generator=g))] = 0          # an empty row
    ns = [0, 1, 2, 3, 5][int(torch.randint(0, 5, (), generator=g))]
    if Hq // Hkv in (1, 2, 4, 8) and torch.rand((), generator=g).item() < 0.5:  # either kernel for these group sizes
        monkeypatch.setitem(ops.DEFAULT_KERNEL, "decode", ["valu", "mfma"][int(torch.randint(0, 2, (), generator=g))])
    case = make_decode_case(2000 + seed, lens, Hq, Hkv, block_size=block, pad_cols=int(torch.randint(0, 3, (), generator=g)))
    d = to_dev(case, gpu)
    order = None
    if torch.rand((), generator=g).item() < 0.5:
        order = torch.tensor(sorted(range(B), key=lambda j: -lens[j]), dtype=torch.int32, device=gpu)
    # a third of the cases run from a host-built work list instead (pieces of a random size, a random
    # cap on pieces per sequence, spare capacity); drawn from a generator of its own, so that the
    # cases above stay what they were before work lists existed
    g2 = torch.Generator().manual_seed(5000 + seed)
    plan = None
    if torch.rand((), generator=g2).item() < 0.34:
        from light_vllm_amd.decode_plan import build_plan
        plan = build_plan(lens, Hkv, Hq // Hkv,
                          max_splits=[1, 3, 64][int(torch.randint(0, 3, (), generator=g2))],
                          piece_tokens=[0, 64, 128, 320][int(torch.randint(0, 4, (), generator=g2))],
                          capacity=[0, 4 * B + 7][int(torch.randint(0, 2, (), generator=g2))]).to(gpu)
    out = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"],
                               num_splits=ns, seq_order=order, plan=plan)
    torch.cuda.synchronize()
    ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= TOL, f"seed {seed}: Hq/Hkv {Hq}/{Hkv} block {block} B {B} splits {ns}: max-abs {err}"
    for b, L in enumerate(lens):
        if L == 0:
            assert bool((out[b] == 0).all())


@pytest.mark.parametrize("seed", range(int(os.environ.get("LV_FUZZ_PREFILL", "20"))))
def test_fuzz_prefill(gpu, monkeypatch, seed):
    from light_vllm_amd import ops
    g = torch.Generator().manual_seed(3000 + seed)
    Hq, Hkv = [(2, 1), (4, 1), (8, 1), (8, 4), (16, 8), (32, 8), (12, 4), (3, 1)][int(torch.randint(0, 8, (), generator=g))]
    B = int(torch.randint(1, 6, (), generator=g))
    lens_q = _lengths(g, B, 600)
    paged = torch.rand((), generator=g).item() < 0.5
    block = [64, 128, 256][int(torch.randint(0, 3, (), generator=g))]
    if paged:   # prefix-cached prefill: the cached part is a whole number of blocks
        lens_k = [lq + block * int(torch.randint(0, 3, (), generator=g)) for lq in lens_q]
    else:
        lens_k = list(lens_q)
    cfg = ["auto", "0", "1", "2", "4"][int(torch.randint(0, 5, (), generator=g))]
    monkeypatch.setitem(ops.DEFAULT_KERNEL, "prefill", {"auto": "auto", "0": "w4", "1": "w8", "2": "pp", "4": "w64"}[cfg])
    # KV runs: the library's choice, none, or a forced count (also more runs than some sequences have key tiles)
    ns = [0, 0, 1, 2, 3, 5, 8][int(torch.randint(0, 7, (), generator=g))]
    case = make_prefill_case(4000 + seed, lens_q, lens_k, Hq, Hkv, block_size=block, paged=paged)
    d = to_dev(case, gpu)
    # four cases in ten launch from a host-built tile table (forced for every head count); with the
    # long-prompt kernel in one pass, half of those from an item table (its persistent form)
    g2 = torch.Generator().manual_seed(6000 + seed)
    tile_order = None
    if torch.rand((), generator=g2).item() < 0.4:
        persistent = cfg == "2" and ns == 1 and torch.rand((), generator=g2).item() < 0.5
        tile_order = ops.prefill_tile_order(lens_q, lens_k, Hq, Hkv, gpu, num_splits=ns, force=True, paged=paged,
                                            persistent=persistent)
    dropped = ops.STATS["tile_order_dropped"]
    out = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                                block_table=d["block_table"], num_splits=ns, tile_order=tile_order)
    torch.cuda.synchronize()
    assert ops.STATS["tile_order_dropped"] == dropped, "a table planned for this very launch was not used"
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"], case["block_table"])
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= TOL, f"seed {seed}: Hq/Hkv {Hq}/{Hkv} cfg {cfg} runs {ns} paged {paged} block {block} lens {lens_q}/{lens_k}: max-abs {err}"

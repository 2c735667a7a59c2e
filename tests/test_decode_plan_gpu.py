"""lv_decode_attn_planned: paged decode driven by a host-built work list (long sequences cut into
more pieces than short ones). Against the CPU oracle at the decode tolerance; bit-equal to
lv_decode_attn whenever the list describes the same cuts (any order, any group size, spare
capacity); a list replayed from a captured graph follows what the host wrote."""
import pytest
import torch

from oracle.attention_oracle import decode_ref
from tests._cases import make_decode_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _args(d):
    return (d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"])


@pytest.mark.parametrize("kernel", ["valu", "mfma"])
@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (8, 1), (4, 1)])
def test_planned_decode_matches_oracle(gpu, Hq, Hkv, kernel):
    from light_vllm_amd import ops
    from light_vllm_amd.decode_plan import build_plan
    lens = [1, 255, 256, 257, 1024, 31, 3000, 33, 700, 5000, 64, 65, 0, 2048]
    case = make_decode_case(21, [max(n, 1) for n in lens], Hq, Hkv)
    case["context_lens"] = torch.tensor(lens, dtype=torch.int32)
    ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    d = to_dev(case, gpu)
    for max_splits, cap, piece in ((64, 0, 0), (4, len(lens) + 40, 0), (1, 0, 0), (64, 0, 256), (64, 200, 64)):
        plan = build_plan(lens, Hkv, Hq // Hkv, max_splits=max_splits, capacity=cap, piece_tokens=piece).to(gpu)
        out = ops.decode_attention(*_args(d), plan=plan, kernel=kernel)
        torch.cuda.synchronize()
        err = (out.float().cpu() - ref).abs().max().item()
        assert err <= TOL, f"max-abs {err} (max_splits {max_splits}, capacity {cap}, piece {piece})"
        assert out[lens.index(0)].abs().sum().item() == 0
        if piece == 256:
            assert max(plan.seq_splits.tolist()) >= 16      # 5000 tokens in 256-token pieces


@pytest.mark.parametrize("kernel", ["valu", "mfma"])
def test_planned_decode_equals_unplanned_on_the_same_cuts(gpu, kernel):
    """A list that gives every sequence the launch's one split count describes exactly what
    lv_decode_attn does: the results must be the same bits, whatever the order of the items, the
    group size and the spare capacity."""
    import random
    from light_vllm_amd import ops
    from light_vllm_amd.decode_plan import DecodePlan, natural_plan
    lens = [900, 17, 2048, 1, 640, 64, 1500, 333, 2047]
    case = make_decode_case(22, lens, 32, 8)
    d = to_dev(case, gpu)
    for ns in (1, 2, 5):
        want = ops.decode_attention(*_args(d), num_splits=ns, kernel=kernel).clone()
        nat = natural_plan(len(lens), ns, capacity=len(lens) * ns + 7)
        got = ops.decode_attention(*_args(d), plan=nat.to(gpu), kernel=kernel)
        assert torch.equal(got, want), f"natural list, {ns} splits"
        # shuffled items, another group size
        body = [(nat.items[2 + 2 * i], nat.items[3 + 2 * i]) for i in range(nat.used)]
        random.Random(ns).shuffle(body)
        items = [nat.used, 4] + [x for it in body for x in it] + [-1, 0] * (nat.capacity - nat.used)
        shuffled = DecodePlan(items, nat.seq_splits, nat.capacity, nat.max_splits, nat.used, 0).to(gpu)
        out = torch.full_like(want, float("nan"))
        ops.decode_attention(*_args(d), plan=shuffled, kernel=kernel, out=out)
        assert torch.equal(out, want), f"shuffled list, {ns} splits"


def test_planned_decode_bad_items_run_nothing(gpu):
    """Items that do not fit the launch are skipped: no fault, other rows unaffected."""
    from light_vllm_amd import ops
    from light_vllm_amd.decode_plan import DecodePlan
    lens = [700, 300]
    case = make_decode_case(23, lens, 16, 8)
    d = to_dev(case, gpu)
    want = ops.decode_attention(*_args(d), num_splits=1).clone()
    items = [4, 32, 0, 0 | (1 << 16), 7, 0 | (1 << 16), 1, 0 | (1 << 16), 1, 5 | (9 << 16)] + [-1, 0] * 3
    plan = DecodePlan(items, [1, 1], 7, 1, 4, 0).to(gpu)
    out = ops.decode_attention(*_args(d), plan=plan)
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_planned_decode_under_graph_replay(gpu):
    """One captured launch, two different batches: the list, the split counts and the lengths are
    device buffers the host rewrites between replays."""
    from light_vllm_amd import ops
    from light_vllm_amd.decode_plan import build_plan, natural_plan
    Hq, Hkv, B, cap, ws_splits = 32, 8, 8, 8 + 64, 16
    lens_a = [4000, 100, 900, 1, 2500, 64, 700, 3000]
    lens_b = [1000] * 6 + [0, 0]                       # a shorter batch with two padding rows
    case = make_decode_case(24, [4096] * B, Hq, Hkv)
    d = to_dev(case, gpu)
    ctx = torch.zeros(B, dtype=torch.int32, device=gpu)
    items = torch.zeros(2 + 2 * cap, dtype=torch.int32, device=gpu)
    ss = torch.ones(B, dtype=torch.int32, device=gpu)
    out = torch.zeros_like(d["q"])
    ops.reserve_decode_workspace(gpu, ops.decode_workspace_bytes(B, Hq, 128, ws_splits))

    def load(plan, lens):
        items.copy_(torch.tensor(plan.items, dtype=torch.int32))
        ss.copy_(torch.tensor(plan.seq_splits, dtype=torch.int32))
        ctx.copy_(torch.tensor(lens, dtype=torch.int32))

    class _P:
        pass
    static = _P()
    static.items, static.seq_splits, static.capacity, static.max_splits, static.workspace_splits = items, ss, cap, ws_splits, ws_splits
    load(build_plan(lens_a, Hkv, Hq // Hkv, max_splits=ws_splits, capacity=cap), lens_a)
    ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], ctx, d["block_table"], d["scale"], plan=static, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], ctx, d["block_table"], d["scale"], plan=static, out=out)
    for lens, plan in ((lens_a, build_plan(lens_a, Hkv, Hq // Hkv, max_splits=ws_splits, capacity=cap)),
                       (lens_b, natural_plan(B, 1, capacity=cap)),
                       (lens_a, build_plan(lens_a, Hkv, Hq // Hkv, max_splits=4, capacity=cap, piece_tokens=512))):
        load(plan, lens)
        out.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        case["context_lens"] = torch.tensor(lens, dtype=torch.int32)
        ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
        err = (out.float().cpu() - ref).abs().max().item()
        assert err <= TOL, f"max-abs {err}"


def test_planned_decode_outlier_batch_at_full_size(gpu):
    """The bench's outlier batch at its own size (256 sequences of 1..1024 tokens, four of them 16384,
    32q/8kv): a size-independent property - permuting the physical blocks (and the block table with
    them) leaves every output bit in place - plus the long sequences and a few short ones against the
    oracle, and the plan's shape (the long ones cut, the short ones not)."""
    from light_vllm_amd import ops
    from light_vllm_amd.decode_plan import build_plan, worth_planning
    g = torch.Generator().manual_seed(5)
    lens = torch.randint(1, 1025, (256,), generator=g).tolist()
    for j in (3, 77, 130, 201):
        lens[j] = 16384
    Hq, Hkv = 32, 8
    assert worth_planning(lens, Hkv, Hq // Hkv)
    case = make_decode_case(31, lens, Hq, Hkv, extra_blocks=0)
    d = to_dev(case, gpu)
    plan = build_plan(lens, Hkv, Hq // Hkv).to(gpu)
    splits = plan.seq_splits.tolist()
    assert all(splits[j] >= 8 for j in (3, 77, 130, 201)) and sum(1 for s in splits if s > 1) == 4
    out = ops.decode_attention(*_args(d), plan=plan)
    nblk = d["k_cache"].size(0)
    perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(1)).to(gpu)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nblk, device=gpu)
    k2, v2 = d["k_cache"][perm], d["v_cache"][perm]
    bt2 = inv[d["block_table"].long()].to(torch.int32)
    out2 = ops.decode_attention(d["q"], k2, v2, d["context_lens"], bt2, d["scale"], plan=plan)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    rows = [3, 201, 0, 255]
    sub = {k: (v[rows] if k in ("q", "context_lens", "block_table") else v) for k, v in case.items()}
    ref = decode_ref(sub["q"], sub["k_cache"], sub["v_cache"], sub["context_lens"], sub["block_table"], sub["scale"])
    assert (out[rows].float().cpu() - ref).abs().max().item() <= TOL

"""Paged decode attention parity: HIP kernel vs the CPU fp32 oracle.

Tolerance: max-abs <= 2e-2 on bf16 outputs (the bound the task states for attention outputs)."""
import pytest
import torch

from oracle.attention_oracle import decode_ref
from tests._cases import make_decode_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-2


def _check(gpu, case, num_splits=0):
    from light_vllm_amd import ops
    ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"],
                     case["block_table"], case["scale"])
    d = to_dev(case, gpu)
    out = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"],
                               d["block_table"], d["scale"], num_splits=num_splits)
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= TOL, f"max-abs {err}"
    return out


HEADS = [(16, 8), (32, 8), (4, 1), (8, 1), (8, 8), (16, 4)]


@pytest.mark.parametrize("Hq,Hkv", HEADS)
def test_decode_ragged_lengths(gpu, Hq, Hkv):
    lens = [1, 255, 256, 257, 1024, 31, 32, 33, 700]
    _check(gpu, make_decode_case(11, lens, Hq, Hkv))


@pytest.mark.parametrize("num_splits", [1, 2, 3, 8])
def test_decode_split_kv(gpu, num_splits):
    lens = [2048, 1, 513, 1500, 256]
    _check(gpu, make_decode_case(5, lens, 32, 8, pad_cols=2), num_splits=num_splits)


def test_decode_heuristic_splits_long_context(gpu):
    _check(gpu, make_decode_case(6, [8192, 5000], 4, 1))


def test_decode_zero_length_rows(gpu):
    """hipGraph padding rows carry context_len 0 and an all-zero block table row."""
    case = make_decode_case(8, [300, 40], 16, 8)
    case["context_lens"] = torch.tensor([300, 0], dtype=torch.int32)
    case["block_table"][1] = 0
    out = _check(gpu, case)
    assert out[1].abs().sum().item() == 0
    out = _check(gpu, case, num_splits=2)
    assert out[1].abs().sum().item() == 0


def test_decode_spiked_scores(gpu):
    """Forces the running-max rescale: one key far above the rest, placed late in the context."""
    case = make_decode_case(9, [1024, 777], 16, 8)
    bt = case["block_table"]
    for b, t in ((0, 900), (1, 776)):
        blk = int(bt[b, t // 256])
        qh = case["q"][b, 0].float()
        case["k_cache"][blk, t % 256, 0] = (qh * 4.0).to(torch.bfloat16)
    _check(gpu, case)
    _check(gpu, case, num_splits=4)


def test_decode_other_block_size(gpu):
    _check(gpu, make_decode_case(10, [100, 64, 65, 1], 16, 8, block_size=64))


def test_decode_config2_shape_properties(gpu):
    """BASELINE config 2 (Qwen3-0.6B, B=256, ctx=1024) at full size: permuting the physical
    blocks (and the block table with them) must not change the output bit pattern, and a
    sample of rows must match the oracle."""
    from light_vllm_amd import ops
    B, ctx, Hq, Hkv = 256, 1024, 16, 8
    case = make_decode_case(0, [ctx] * B, Hq, Hkv, extra_blocks=0)
    d = to_dev(case, gpu)
    out = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"])
    nblk = d["k_cache"].size(0)
    perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(1)).to(gpu)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nblk, device=gpu)
    k2, v2 = d["k_cache"][perm], d["v_cache"][perm]
    bt2 = inv[d["block_table"].long()].to(torch.int32)
    out2 = ops.decode_attention(d["q"], k2, v2, d["context_lens"], bt2, d["scale"])
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))
    rows = [0, 17, 255]
    sub = {k: (v[rows] if k in ("q", "context_lens", "block_table") else v) for k, v in case.items()}
    ref = decode_ref(sub["q"], sub["k_cache"], sub["v_cache"], sub["context_lens"], sub["block_table"], sub["scale"])
    assert (out[rows].float().cpu() - ref).abs().max().item() <= TOL


# ---- The library picks the matrix-core kernel for 4 and 8 query heads per kv head and the
# ---- vector-ALU kernel for 1 and 2; the kernel hint forces one of them for every group size.
@pytest.mark.parametrize("mfma", ["valu", "mfma"])
@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (64, 8), (16, 2), (4, 1), (32, 8), (16, 8), (8, 8)])
def test_decode_wide_groups_both_kernels(gpu, monkeypatch, Hq, Hkv, mfma):
    from light_vllm_amd import ops
    monkeypatch.setitem(ops.DEFAULT_KERNEL, "decode", mfma)
    lens = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 1024, 700]
    _check(gpu, make_decode_case(12, lens, Hq, Hkv))


@pytest.mark.parametrize("num_splits", [1, 2, 3, 8, 0])
def test_decode_wide_groups_split_kv(gpu, num_splits):
    lens = [2048, 1, 513, 1500, 256]
    _check(gpu, make_decode_case(13, lens, 8, 1, pad_cols=2), num_splits=num_splits)
    _check(gpu, make_decode_case(14, lens, 64, 8, pad_cols=1), num_splits=num_splits)


def test_decode_wide_groups_zero_rows_spikes_block64(gpu):
    case = make_decode_case(15, [300, 40], 64, 8)
    case["context_lens"] = torch.tensor([300, 0], dtype=torch.int32)
    case["block_table"][1] = 0
    out = _check(gpu, case)
    assert out[1].abs().sum().item() == 0
    out = _check(gpu, case, num_splits=2)
    assert out[1].abs().sum().item() == 0
    case = make_decode_case(16, [1024, 777], 8, 1)
    bt = case["block_table"]
    for b, t, hq in ((0, 900, 0), (1, 776, 5), (0, 40, 7)):
        blk = int(bt[b, t // 256])
        case["k_cache"][blk, t % 256, 0] = (case["q"][b, hq].float() * 4.0).to(torch.bfloat16)
    _check(gpu, case)
    _check(gpu, case, num_splits=4)
    _check(gpu, make_decode_case(17, [100, 64, 65, 1], 8, 1, block_size=64))


@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (8, 1)])
def test_decode_seq_order_does_not_change_results(gpu, Hq, Hkv):
    """seq_order only changes which workgroup serves which sequence: outputs are bit-identical for
    the identity, longest-first and a random permutation; an out-of-range entry skips that row."""
    from light_vllm_amd import ops
    lens = [5, 1024, 33, 700, 256, 1, 999, 64]
    case = make_decode_case(18, lens, Hq, Hkv)
    d = to_dev(case, gpu)
    args = (d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"])
    base = ops.decode_attention(*args)
    B = len(lens)
    longest_first = sorted(range(B), key=lambda j: -lens[j])
    g = torch.Generator().manual_seed(3)
    for order in (list(range(B)), longest_first, torch.randperm(B, generator=g).tolist()):
        so = torch.tensor(order, dtype=torch.int32, device=gpu)
        for ns in (0, 1, 3):
            out = ops.decode_attention(*args, num_splits=ns, seq_order=so)
            ref = base if ns == 0 else ops.decode_attention(*args, num_splits=ns)
            assert torch.equal(out, ref)
    # a negative first entry = no order at all (how a captured graph switches the order off)
    so = torch.tensor([-1, 7, 7, 7, 99, -5, 0, 0], dtype=torch.int32, device=gpu)
    assert torch.equal(ops.decode_attention(*args, num_splits=1, seq_order=so), ops.decode_attention(*args, num_splits=1))
    so = torch.tensor([0, 1, 2, 3, 4, 5, 6, 99], dtype=torch.int32, device=gpu)
    out = torch.full_like(base, 7.0)
    ops.decode_attention(*args, num_splits=1, out=out, seq_order=so)
    torch.cuda.synchronize()
    ref = ops.decode_attention(*args, num_splits=1)
    assert torch.equal(out[:7], ref[:7]) and bool((out[7] == 7.0).all())


@pytest.mark.parametrize("B", [32, 37, 70, 96])
@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (4, 1)])
def test_decode_seq_order_grouped_batches(gpu, B, Hq, Hkv):
    """With an order the workgroups walk groups of 32 launch positions (all kv heads of a group
    before the next group); the last group may be short. Every (sequence, kv head) must be served
    exactly once: bit-identical to the run without an order, for 1 and 3 splits."""
    from light_vllm_amd import ops
    g = torch.Generator().manual_seed(B)
    lens = torch.randint(1, 600, (B,), generator=g).tolist()
    case = make_decode_case(19, lens, Hq, Hkv)
    d = to_dev(case, gpu)
    args = (d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"])
    longest_first = sorted(range(B), key=lambda j: -lens[j])
    so = torch.tensor(longest_first, dtype=torch.int32, device=gpu)
    for ns in (1, 3):
        ref = ops.decode_attention(*args, num_splits=ns)
        out = torch.full_like(ref, 7.0)
        ops.decode_attention(*args, num_splits=ns, out=out, seq_order=so)
        assert torch.equal(out, ref)
    _check(gpu, case)


@pytest.mark.parametrize("Hq,Hkv", [(40, 8), (5, 1), (3, 1), (6, 1), (7, 1), (12, 2)])
def test_decode_other_group_sizes(gpu, Hq, Hkv):
    """Group sizes 3, 5, 6, 7 (Qwen3-14B: 40 q / 8 kv heads) run on the matrix-core kernel."""
    case = make_decode_case(23, [5, 300, 257, 64, 1000], Hq, Hkv)
    _check(gpu, case)
    _check(gpu, case, num_splits=3)

"""Varlen causal prefill parity: HIP MFMA kernel vs the CPU fp32 oracle (max-abs <= 2e-2)."""
import pytest
import torch

from oracle.attention_oracle import prefill_ref
from tests._cases import make_prefill_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-2


@pytest.fixture(autouse=True, params=["auto", "0", "1", "2", "4"], ids=lambda c: f"cfg-{c}")
def kernel_config(request, monkeypatch):
    """Every case runs under the library's own choice of kernel and under each forced one:
    0 = 4 waves / 128-row tiles, 1 = 8 waves / 256-row tiles, 2 = the phase-locked 8-wave kernel."""
    from light_vllm_amd import ops
    monkeypatch.setitem(ops.DEFAULT_KERNEL, "prefill",
                        {"auto": "auto", "0": "w4", "1": "w8", "2": "pp", "4": "w64"}[request.param])
    return request.param


def _run(gpu, case, num_splits=0):
    from light_vllm_amd import ops
    d = to_dev(case, gpu)
    out = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"],
                                d["scale"], block_table=d["block_table"], num_splits=num_splits)
    torch.cuda.synchronize()
    return out


def _check(gpu, case, num_splits=0):
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"],
                      case["block_table"])
    out = _run(gpu, case, num_splits)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err <= TOL, f"max-abs {err}"
    return out


HEADS = [(16, 8), (32, 8), (4, 1), (8, 1)]


@pytest.mark.parametrize("Hq,Hkv", HEADS)
def test_prefill_packed_ragged(gpu, Hq, Hkv):
    lens = [1, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 700]
    _check(gpu, make_prefill_case(21, lens, lens, Hq, Hkv))


def test_prefill_single_long(gpu):
    _check(gpu, make_prefill_case(22, [1500], [1500], 16, 8))


@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (8, 1)])
def test_prefill_paged_prefix(gpu, Hq, Hkv):
    """Prefix-cached prefill: Lq = len - cached, Lk = len, K/V read through the block table."""
    lens_k = [600, 256 + 17, 1024, 300]
    cached = [256, 256, 768, 0]
    lens_q = [k - c for k, c in zip(lens_k, cached)]
    _check(gpu, make_prefill_case(23, lens_q, lens_k, Hq, Hkv, paged=True))


def test_prefill_paged_block64(gpu):
    _check(gpu, make_prefill_case(24, [100, 7], [228, 71], 16, 8, block_size=64, paged=True))


def test_prefill_strided_inputs(gpu):
    """q/k/v as views into a fused qkv activation (row stride (Hq+2Hkv)*D), as models/qwen3.py:95."""
    from light_vllm_amd import ops
    Hq, Hkv, D = 16, 8, 128
    case = make_prefill_case(25, [130, 64], [130, 64], Hq, Hkv)
    N = case["q"].size(0)
    qkv = torch.empty(N, (Hq + 2 * Hkv) * D, dtype=torch.bfloat16)
    qkv[:, :Hq * D] = case["q"].view(N, -1)
    qkv[:, Hq * D:(Hq + Hkv) * D] = case["k"].view(N, -1)
    qkv[:, (Hq + Hkv) * D:] = case["v"].view(N, -1)
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"])
    g = qkv.to(gpu)
    q, k, v = g.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    out = ops.prefill_attention(q.view(N, Hq, D), k.view(N, Hkv, D), v.view(N, Hkv, D),
                                case["cu_q"].to(gpu), case["cu_k"].to(gpu), 130, 130, case["scale"])
    torch.cuda.synchronize()
    assert (out.float().cpu() - ref).abs().max().item() <= TOL


def test_prefill_spiked_scores(gpu):
    """Forces the running-max rescale: a late key aligned with one query row, far above the rest."""
    case = make_prefill_case(26, [512], [512], 16, 8)
    case["k"][400, 0] = (case["q"][450, 1].float() * 4.0).to(torch.bfloat16)
    case["k"][70, 3] = (case["q"][300, 6].float() * 4.0).to(torch.bfloat16)
    _check(gpu, case)


def test_prefill_config3_shape_sampled(gpu):
    """BASELINE config 3 (Qwen3-8B, 4x4096 packed) at full size: row blocks sampled against the
    oracle (a causal row block r0..r1 of a sequence only depends on keys 0..r1)."""
    Hq, Hkv, L = 32, 8, 4096
    case = make_prefill_case(27, [L] * 4, [L] * 4, Hq, Hkv)
    out = _run(gpu, case).float().cpu()
    for b, r0, r1 in [(0, 0, 40), (1, 2000, 2040), (3, 4056, 4096), (2, 120, 136)]:
        s = b * L
        q = case["q"][s + r0:s + r1]
        k = case["k"][s:s + r1]
        v = case["v"][s:s + r1]
        cu_q = torch.tensor([0, r1 - r0], dtype=torch.int32)
        cu_k = torch.tensor([0, r1], dtype=torch.int32)
        ref = prefill_ref(q, k, v, cu_q, cu_k, case["scale"])
        err = (out[s + r0:s + r1] - ref).abs().max().item()
        assert err <= TOL, f"seq {b} rows {r0}:{r1} max-abs {err}"


def test_prefill_matches_decode_last_row(gpu):
    """Size-independent property: the last prefill row of a sequence equals decode over its cache."""
    from light_vllm_amd import ops
    Hq, Hkv, D, L = 32, 8, 128, 777
    case = make_prefill_case(28, [L - 512], [L], Hq, Hkv, paged=True)
    d = to_dev(case, gpu)
    out = _run(gpu, case)
    q_last = d["q"][-1:].contiguous()
    dec = ops.decode_attention(q_last, d["k"], d["v"], torch.tensor([L], dtype=torch.int32, device=gpu),
                               d["block_table"], d["scale"])
    torch.cuda.synchronize()
    assert (out[-1:].float() - dec.float()).abs().max().item() <= 1e-2


def test_prefill_odd_tile_counts(gpu):
    """Tile counts 1..5 per 256-row query tile, partial last tiles, and a sequence longer than 2048
    (the length from which the library picks the phase-locked kernel on its own)."""
    lens = [64, 65, 128, 129, 192, 257, 320, 2049 + 77]
    _check(gpu, make_prefill_case(29, lens, lens, 16, 8))


def test_prefill_spiked_scores_long(gpu):
    """Running-max rescale late in a long sequence, in both 32-key halves of a tile and in the
    last tile of a row block."""
    L = 2304
    case = make_prefill_case(30, [L], [L], 8, 1)
    case["k"][2000, 0] = (case["q"][2100, 1].float() * 4.0).to(torch.bfloat16)
    case["k"][2000 + 40, 0] = (case["q"][2200, 5].float() * 5.0).to(torch.bfloat16)
    case["k"][2303, 0] = (case["q"][2303, 7].float() * 6.0).to(torch.bfloat16)
    _check(gpu, case)


# ---- KV splits of the query tiles (few query rows against a long context). Forced counts run
# ---- under every kernel hint of the fixture; the two long-prompt kernels hand over to the 8-wave one.
@pytest.mark.parametrize("num_splits", [2, 5, 32])
def test_prefill_split_kv_packed_ragged(gpu, num_splits):
    """Ragged packed batch: tile counts below, at and above the split count, so some runs are empty."""
    lens = [1, 63, 64, 65, 300, 1000, 2100]
    _check(gpu, make_prefill_case(31, lens, lens, 16, 8), num_splits=num_splits)


@pytest.mark.parametrize("Hq,Hkv", [(32, 8), (8, 1)])
@pytest.mark.parametrize("num_splits", [0, 3, 8])
def test_prefill_split_kv_prefix_cached(gpu, Hq, Hkv, num_splits):
    """The shape the splits are for: a few new tokens per sequence on top of a long cached prefix,
    read through the block table (BASELINE config 5's prefix-cache hits). 0 = the heuristic, which
    splits here (2 or 3 query tiles x heads against 36-70 key tiles)."""
    from light_vllm_amd import ops
    lens_k = [2304 + 40, 4352 + 100, 1024 + 7]
    cached = [2304, 4352, 1024]
    lens_q = [k - c for k, c in zip(lens_k, cached)]
    if num_splits == 0:
        assert ops._lib.load().lv_prefill_num_splits(3, max(lens_q), max(lens_k), Hq) > 1
    _check(gpu, make_prefill_case(32, lens_q, lens_k, Hq, Hkv, paged=True), num_splits=num_splits)


def test_prefill_split_kv_spiked_and_bit_stable(gpu):
    """A run that holds a score far above the others must dominate the merge; and the split path is
    deterministic (two launches, same bits)."""
    case = make_prefill_case(33, [200], [200 + 1792], 16, 8, paged=True)
    bt = case["block_table"]
    t = 1900
    case["k"][int(bt[0, t // 256]), t % 256, 0] = (case["q"][150, 1].float() * 4.0).to(torch.bfloat16)
    a = _check(gpu, case, num_splits=6)
    b = _run(gpu, case, num_splits=6)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))


@pytest.mark.parametrize("num_splits", [0, 1, 4])
def test_prefill_sequence_without_new_tokens(gpu, num_splits):
    """A fully cached prompt leaves its sequence with no query rows (cu_seqlens_q repeats a value;
    the reference does not special-case it either, engine/model_runner.py:318-328). The launch
    must neither read nor write anything for it and the other sequences must come out as always,
    whether it sits first, in the middle or last, and with KV runs."""
    lens_k = [512, 600, 256, 1024, 768]
    cached = [512, 256, 256, 768, 768]
    lens_q = [k - c for k, c in zip(lens_k, cached)]          # [0, 344, 0, 256, 0]
    case = make_prefill_case(61, lens_q, lens_k, 32, 8, paged=True)
    d = to_dev(case, gpu)
    from light_vllm_amd import ops
    out = torch.full((sum(lens_q) + 8, 32, 128), 7.0, dtype=torch.bfloat16, device=gpu)   # 8 canary rows behind the output
    ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                          block_table=d["block_table"], num_splits=num_splits, out=out[:sum(lens_q)])
    torch.cuda.synchronize()
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"], case["block_table"])
    assert (out[:sum(lens_q)].float().cpu() - ref).abs().max().item() <= TOL
    assert bool((out[sum(lens_q):] == 7.0).all())


# ---- host-built launch order of the query tiles: the same bits as the library's own mapping
def _order_for(gpu, lens_q, lens_k, Hq, Hkv, num_splits=0):
    from light_vllm_amd import ops
    return ops.prefill_tile_order(lens_q, lens_k, Hq, Hkv, gpu, num_splits=num_splits, force=True)


@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (8, 2), (16, 4), (32, 8), (6, 2), (9, 3)])
def test_prefill_tile_order_same_bits(gpu, Hq, Hkv):
    """Packed ragged batch incl. an empty sequence: every tile appears once in the table, so the
    output is the library's own, bit for bit; rows behind the output stay untouched."""
    from light_vllm_amd import ops
    lens = [700, 1, 0, 129, 2100, 256, 513]
    case = make_prefill_case(71, lens, lens, Hq, Hkv)
    d = to_dev(case, gpu)
    order = _order_for(gpu, lens, lens, Hq, Hkv)
    _check(gpu, case)
    # the plan may name a kernel of its own (128-row tiles for a small grid): compare like with like
    base = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                                 kernel=order.kernel)
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"])
    assert (base.float().cpu() - ref).abs().max().item() <= TOL
    dropped = ops.STATS["tile_order_dropped"]
    out = torch.full((sum(lens) + 8, Hq, 128), 7.0, dtype=torch.bfloat16, device=gpu)
    ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                          out=out[:sum(lens)], tile_order=order)
    torch.cuda.synchronize()
    assert ops.STATS["tile_order_dropped"] == dropped, "the table did not match the launch it was built for"
    assert torch.equal(out[:sum(lens)].view(torch.int16), base.view(torch.int16))
    assert bool((out[sum(lens):] == 7.0).all())


@pytest.mark.parametrize("Hq,Hkv", [(8, 1), (32, 8)])
@pytest.mark.parametrize("num_splits", [0, 3])
def test_prefill_tile_order_prefix_cached_splits(gpu, Hq, Hkv, num_splits):
    """Paged prefix mode with KV runs: table and split dimension together."""
    from light_vllm_amd import ops
    lens_k = [2304 + 40, 4352 + 300, 1024 + 7, 512]
    cached = [2304, 4352, 1024, 512]
    lens_q = [k - c for k, c in zip(lens_k, cached)]
    case = make_prefill_case(72, lens_q, lens_k, Hq, Hkv, paged=True)
    d = to_dev(case, gpu)
    order = _order_for(gpu, lens_q, lens_k, Hq, Hkv, num_splits)
    base = _check(gpu, case, num_splits=num_splits)
    dropped = ops.STATS["tile_order_dropped"]
    out = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                                block_table=d["block_table"], num_splits=num_splits, tile_order=order)
    torch.cuda.synchronize()
    assert ops.STATS["tile_order_dropped"] == dropped
    assert torch.equal(out.view(torch.int16), base.view(torch.int16))


def test_prefill_tile_order_for_other_tile_height_is_rejected_by_the_abi(gpu):
    """ops drops a table built for another kernel; the C entry point itself refuses it."""
    from light_vllm_amd import ops, _lib
    lens = [300, 200]
    case = make_prefill_case(73, lens, lens, 8, 1)
    d = to_dev(case, gpu)
    from light_vllm_amd.prefill_order import build_tile_order
    hint = ops.PREFILL_KERNELS[ops.DEFAULT_KERNEL["prefill"]]
    rows = _lib.load().lv_prefill_tile_rows(300, 1, hint)
    wrong = build_tile_order(lens, lens, 300, 384 - rows, 1).to(gpu)      # 256 <-> 128
    before = ops.STATS["tile_order_dropped"]
    out = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], 300, 300, d["scale"], tile_order=wrong)
    torch.cuda.synchronize()
    assert ops.STATS["tile_order_dropped"] == before + 1
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"])
    assert (out.float().cpu() - ref).abs().max().item() <= TOL
    lib = _lib.load()
    q, k, v = d["q"], d["k"], d["v"]
    rc = lib.lv_prefill_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(0), k.stride(0),
                             v.stride(0), out.stride(0), d["cu_q"].data_ptr(), d["cu_k"].data_ptr(), 2, 300, 300,
                             None, 0, 0, 8, 1, 128, float(d["scale"]), q.size(0), None, 0, 1,
                             wrong.table.data_ptr(), wrong.table.numel(), wrong.tile_rows, 0, hint, None)
    assert rc == -1 and b"tile_order" in lib.lv_last_error()
    rc = lib.lv_prefill_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q.stride(0), k.stride(0),
                             v.stride(0), out.stride(0), d["cu_q"].data_ptr(), d["cu_k"].data_ptr(), 2, 300, 300,
                             None, 0, 0, 8, 1, 128, float(d["scale"]), q.size(0), None, 0, 1,
                             wrong.table.data_ptr(), 7, rows, 0, hint, None)
    assert rc == -1 and b"tile_order_len" in lib.lv_last_error()


def test_prefill_tile_order_planned_on_smaller_tiles(gpu, kernel_config):
    """A launch of at most one 256-row workgroup per CU is planned on 128-row tiles (the table then
    names its kernel); the output is still the default launch's."""
    from light_vllm_amd import ops
    if kernel_config != "auto":
        pytest.skip("the plan only overrides the library's own kernel choice")
    lens = [2100, 2048]            # 9 + 8 tiles of 256 rows x 8 heads = 136 workgroups
    case = make_prefill_case(74, lens, lens, 8, 2)
    d = to_dev(case, gpu)
    order = ops.prefill_tile_order(lens, lens, 8, 2, gpu)
    assert order is not None and order.kernel == "w4" and order.tile_rows == 128 and order.cols == 4
    base = _check(gpu, case)
    out = ops.prefill_attention(d["q"], d["k"], d["v"], d["cu_q"], d["cu_k"], d["max_q"], d["max_k"], d["scale"],
                                tile_order=order)
    torch.cuda.synchronize()
    ref = prefill_ref(case["q"], case["k"], case["v"], case["cu_q"], case["cu_k"], case["scale"])
    assert (out.float().cpu() - ref).abs().max().item() <= TOL
    assert (out.float() - base.float()).abs().max().item() <= 1e-2      # another kernel: same function
    assert ops.prefill_tile_order(lens, lens, 32, 8, gpu) is None        # 8 kv heads: the library's mapping

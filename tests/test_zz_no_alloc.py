"""lv_decode_attn_static, the decode entry point that allocates nothing: the split-KV partials go
to a device array built into the library. It must return what lv_decode_attn returns, bit for bit,
on back-to-back calls that share the array (the second one smaller, so stale partials of the first
lie behind it), must not move the allocator, and must reject sizes beyond the array."""

import pytest
import torch

from oracle.attention_oracle import decode_ref
from tests._cases import make_decode_case, to_dev

pytestmark = pytest.mark.gpu
TOL = 2e-2
SCRATCH_BYTES = 17 << 20


def _call(lib, fn_static, d, out, nsplit, ws=None, hint=0):
    q, kc, vc, ctx, bt = d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"]
    B, Hq, D = q.shape
    common = (q.data_ptr(), kc.data_ptr(), vc.data_ptr(), out.data_ptr(), q.stride(0), out.stride(0),
              ctx.data_ptr(), bt.data_ptr(), bt.stride(0), B, bt.size(1), kc.size(1), Hq, kc.size(2), D,
              float(d["scale"]))
    stream = torch.cuda.current_stream().cuda_stream
    if fn_static:
        return lib.lv_decode_attn_static(*common, nsplit, None, hint, stream)
    return lib.lv_decode_attn(*common, ws.data_ptr() if ws is not None else None,
                              ws.numel() if ws is not None else 0, nsplit, None, hint, stream)


@pytest.mark.parametrize("Hq,Hkv", [(16, 8), (32, 8), (4, 1)])
def test_static_scratch_matches_workspace_entry_on_two_calls(gpu, Hq, Hkv):
    from light_vllm_amd import _lib
    lib = _lib.load()
    big = to_dev(make_decode_case(71, [2048, 1, 700, 1500, 256, 3000, 64, 65], Hq, Hkv), gpu)
    small = to_dev(make_decode_case(72, [513, 300, 1], Hq, Hkv), gpu)
    ws = torch.empty(SCRATCH_BYTES, dtype=torch.uint8, device=gpu)
    outs = {}
    for name, d, nsplit in (("big", big, 8), ("small", small, 3)):
        want = torch.full_like(d["q"], float("nan"))
        assert _call(lib, False, d, want, nsplit, ws) == 0
        outs[name] = (d, want, nsplit)
    torch.cuda.synchronize()
    # one call first so the array's address is resolved, then measure: no allocation of any kind
    first = torch.empty_like(big["q"])
    assert _call(lib, True, big, first, 8) == 0, lib.lv_last_error()
    torch.cuda.synchronize()
    got_big = torch.full_like(big["q"], float("nan"))
    got_small = torch.full_like(small["q"], float("nan"))
    free0, _ = torch.cuda.mem_get_info()
    alloc0 = torch.cuda.memory_stats()["allocation.all.allocated"]
    assert _call(lib, True, big, got_big, 8) == 0, lib.lv_last_error()
    assert _call(lib, True, small, got_small, 3) == 0, lib.lv_last_error()     # second call: smaller input
    torch.cuda.synchronize()
    assert torch.cuda.memory_stats()["allocation.all.allocated"] == alloc0
    assert torch.cuda.mem_get_info()[0] == free0
    for got, (d, want, _) in ((got_big, outs["big"]), (got_small, outs["small"])):
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert torch.equal(first.view(torch.int16), outs["big"][1].view(torch.int16))
    case = make_decode_case(72, [513, 300, 1], Hq, Hkv)
    ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    assert (got_small.float().cpu() - ref).abs().max().item() <= TOL


def test_static_scratch_heuristic_splits_and_single_pass(gpu):
    """num_splits 0 (library heuristic) and 1 (no scratch use at all) through the same entry."""
    from light_vllm_amd import _lib, ops
    lib = _lib.load()
    d = to_dev(make_decode_case(73, [4000, 2500, 9], 8, 2), gpu)
    for nsplit in (0, 1):
        want = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"],
                                    d["scale"], num_splits=nsplit)
        got = torch.empty_like(want)
        assert _call(lib, True, d, got, nsplit) == 0, lib.lv_last_error()
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    got2 = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"],
                                out=torch.empty_like(d["q"]), static_scratch=True)
    want0 = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"])
    torch.cuda.synchronize()
    assert torch.equal(got2.view(torch.int16), want0.view(torch.int16))


def test_static_scratch_rejects_sizes_beyond_the_array(gpu):
    from light_vllm_amd import _lib
    lib = _lib.load()
    # 64 sequences x 32 heads x 512 splits x 520 B = 545 MB of partials > 17 MiB: refused before any launch
    d = to_dev(make_decode_case(74, [64] * 64, 32, 8), gpu)
    out = torch.zeros_like(d["q"])
    rc = _call(lib, True, d, out, 512)
    assert rc == -4 and b"scratch" in lib.lv_last_error()
    torch.cuda.synchronize()
    assert not out.any()
    assert lib.lv_decode_workspace_bytes(64, 32, 128, 512) > SCRATCH_BYTES

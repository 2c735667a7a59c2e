"""Decode parity on LARGE score magnitudes (q scaled 3x / 6x, keys growing with the block index):
both kernels, split and unsplit, contexts up to 3000 tokens. Bound per element: 2e-2 plus one bf16
rounding step of the reference value."""
import pytest
import torch

from oracle.attention_oracle import decode_ref
from tests._cases import make_decode_case, to_dev
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(21))
def test_decode_spiky(gpu, monkeypatch, seed):
    from light_vllm_amd import ops
    g = torch.Generator().manual_seed(9000 + seed)
    Hq, Hkv = [(4, 1), (16, 8), (8, 8), (5, 1), (32, 8), (2, 1), (8, 1)][seed % 7]
    B = int(torch.randint(1, 9, (), generator=g))
    lens = [int(torch.randint(1, 3001, (), generator=g)) for _ in range(B)]
    ns = [0, 1, 2, 5][int(torch.randint(0, 4, (), generator=g))]
    if Hq // Hkv in (1, 2, 4, 8):
        monkeypatch.setitem(ops.DEFAULT_KERNEL, "decode", ["valu", "mfma"][seed % 2])
    case = make_decode_case(9500 + seed, lens, Hq, Hkv)
    case["q"] = (case["q"].float() * [1.0, 3.0, 6.0][seed % 3]).to(torch.bfloat16)
    if seed % 2:
        ramp = torch.linspace(1.0, 3.0, case["k_cache"].shape[0] * 256).view(-1, 256, 1, 1)
        case["k_cache"] = (case["k_cache"].float() * ramp).to(torch.bfloat16)
    d = to_dev(case, gpu)
    out = ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"], d["scale"], num_splits=ns)
    torch.cuda.synchronize()
    ref = decode_ref(case["q"], case["k_cache"], case["v_cache"], case["context_lens"], case["block_table"], case["scale"])
    excess = ((out.float().cpu() - ref).abs() - (2e-2 + ref.abs() * 2.0 ** -7)).max().item()
    assert excess <= 0, f"seed {seed} Hq/Hkv {Hq}/{Hkv} ns {ns} lens {lens}: {excess}"

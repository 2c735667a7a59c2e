"""HIP kernels for the memory-bound steps around attention (lv_qk_norm_rope_store, lv_add_rms_norm,
lv_silu_and_mul) against the CPU layer oracle and the committed outputs of the reference's own
modules. Bar: cache rows bit-identical to the q/k/v rows they are copies of; at least 99 % of all
computed elements bit-identical to the oracle; the rest within TWO bf16 rounding steps. (The mean
of squares is summed in a different order and rsqrt is the device's, so x * rsqrt(...) can land on
the other side of a rounding boundary: one step, which the following multiplication by the weight
can stretch to two when it crosses into a binade with half the relative spacing.) After the
rotation such a difference in a normalised value n shows up as at most step(n) * (|cos| + |sin|) in
either output of its pair, which can be many steps of an output that cancelled to something small,
so rotated values are held to 3 steps of the larger member of their input pair instead."""
import pytest
import torch

from oracle import layers_oracle as lo
from tests._cases import rand_bf16
from tests.test_oracle_cpu import _layers_golden

pytestmark = pytest.mark.gpu
D = 128
EPS = 1e-6


def _ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 rounding steps (sign-magnitude bit patterns mapped to a number line)."""
    def line(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (line(a.cpu()) - line(b.cpu())).abs()


def _close(got, want, what):
    d = _ulp_diff(got, want)
    assert int(d.max()) <= 2, f"{what}: {int(d.max())} bf16 steps off"
    frac = float((d == 0).float().mean())
    assert frac >= 0.99, f"{what}: only {frac:.4f} of elements bit-identical"


def _close_rotated(got, want, normed, what):
    """normed: the oracle's values before the rotation, [N, H, 128]."""
    mag = normed.float().abs()
    pair = torch.maximum(mag[..., :64], mag[..., 64:])
    tol = 3 * 2.0 ** -7 * torch.cat((pair, pair), dim=-1) + 1e-30
    diff = (got.float().cpu() - want.float()).abs()
    assert bool((diff <= tol).all()), f"{what}: off by {float((diff / tol).max()):.2f} x tolerance"
    frac = float((_ulp_diff(got, want) == 0).float().mean())
    assert frac >= 0.99, f"{what}: only {frac:.4f} of elements bit-identical"


def _bits_equal(a, b):
    return torch.equal(a.cpu().contiguous().view(torch.int16), b.cpu().contiguous().view(torch.int16))


def test_golden_from_reference_modules(gpu):
    from light_vllm_amd import ops
    g, meta = _layers_golden()
    eps = float(meta["eps"])
    table = lo.cos_sin_table(D, int(meta["max_position"]), float(meta["base"])).to(gpu)
    q, k = g["q"].to(gpu), g["k"].to(gpu)
    v = torch.zeros_like(k)
    ops.qk_norm_rope_store(q, k, v, g["positions"].to(gpu), g["q_weight"].to(gpu), g["k_weight"].to(gpu),
                           eps, table)
    _close_rotated(q, g["q_rot"], g["q_normed"], "q")
    _close_rotated(k, g["k_rot"], g["k_normed"], "k")
    out = ops.add_rms_norm(g["x"].to(gpu), None, g["weight"].to(gpu), eps)
    _close(out, g["rms_out"], "rms")
    res = g["residual"].to(gpu)
    out = ops.add_rms_norm(g["x"].to(gpu), res, g["weight"].to(gpu), eps)
    _close(out, g["add_out"], "add_rms out")
    assert _bits_equal(res, g["add_res"])      # a single fp32 add + rounding: exact


@pytest.mark.parametrize("N,Hq,Hkv", [(1, 16, 8), (7, 32, 8), (256, 32, 8), (300, 4, 1), (1031, 8, 1), (64, 64, 8)])
def test_qk_norm_rope_store(gpu, N, Hq, Hkv):
    """q/k/v as strided views of one fused qkv activation, as Qwen3Attention produces them."""
    from light_vllm_amd import ops
    gen = torch.Generator().manual_seed(1000 + N)
    max_pos = 4096
    qkv = rand_bf16(gen, N, (Hq + 2 * Hkv) * D, scale=2.0)
    wq = (1 + 0.3 * torch.randn(D, generator=gen)).to(torch.bfloat16)
    wk = (1 + 0.3 * torch.randn(D, generator=gen)).to(torch.bfloat16)
    positions = torch.randint(0, max_pos, (N,), generator=gen, dtype=torch.int64)
    table = lo.cos_sin_table(D, max_pos, 1e6)
    num_slots = 4 * 256
    slots = torch.randperm(num_slots, generator=gen)[:N].to(torch.int32) if N <= num_slots else None
    if slots is None:
        num_slots = 8 * 256
        slots = torch.randperm(num_slots, generator=gen)[:N].to(torch.int32)
    slots[N // 2] = -1                                       # padding row: nothing stored
    kc0 = rand_bf16(gen, num_slots // 256, 256, Hkv, D)
    vc0 = rand_bf16(gen, num_slots // 256, 256, Hkv, D)

    q_ref, k_ref, v_ref = qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    q_want, k_want = lo.qk_norm_rope_ref(q_ref.reshape(N, Hq, D), k_ref.reshape(N, Hkv, D), positions,
                                         wq, wk, EPS, table)

    d_qkv = qkv.to(gpu)
    dq, dk, dv = d_qkv.split([Hq * D, Hkv * D, Hkv * D], dim=-1)
    kc, vc = kc0.to(gpu), vc0.to(gpu)
    ops.qk_norm_rope_store(dq.view(N, Hq, D), dk.view(N, Hkv, D), dv.view(N, Hkv, D), positions.to(gpu),
                           wq.to(gpu), wk.to(gpu), EPS, table.to(gpu), kc, vc, slots.to(gpu))
    torch.cuda.synchronize()
    _close_rotated(dq.reshape(N, Hq, D), q_want, lo.rms_norm_ref(q_ref.reshape(N, Hq, D), wq, EPS), "q")
    _close_rotated(dk.reshape(N, Hkv, D), k_want, lo.rms_norm_ref(k_ref.reshape(N, Hkv, D), wk, EPS), "k")
    assert _bits_equal(dv, v_ref)                            # v is read-only
    # cache rows: exact copies of what the kernel left in k / found in v; other rows untouched
    kc_want, vc_want = kc0.clone().view(-1, Hkv * D), vc0.clone().view(-1, Hkv * D)
    keep = slots >= 0
    kc_want[slots[keep].long()] = dk.cpu().reshape(N, Hkv * D)[keep]
    vc_want[slots[keep].long()] = v_ref.reshape(N, Hkv * D)[keep]
    assert _bits_equal(kc.view(-1, Hkv * D), kc_want)
    assert _bits_equal(vc.view(-1, Hkv * D), vc_want)


def test_qk_norm_rope_without_store_and_position_clamp(gpu):
    from light_vllm_amd import ops
    gen = torch.Generator().manual_seed(5)
    N, Hq, Hkv, max_pos = 9, 16, 8, 64
    q, k, v = rand_bf16(gen, N, Hq, D), rand_bf16(gen, N, Hkv, D), rand_bf16(gen, N, Hkv, D)
    w = torch.ones(D, dtype=torch.bfloat16)
    positions = torch.tensor([0, 63, 64, 1000, -5, 1, 2, 3, 4], dtype=torch.int64)
    table = lo.cos_sin_table(D, max_pos, 1e6)
    q_want, k_want = lo.qk_norm_rope_ref(q, k, positions.clamp(0, max_pos - 1), w, w, EPS, table)
    dq, dk = q.to(gpu), k.to(gpu)
    ops.qk_norm_rope_store(dq, dk, v.to(gpu), positions.to(gpu), w.to(gpu), w.to(gpu), EPS, table.to(gpu),
                           torch.tensor([]), torch.tensor([]), None)     # unallocated caches: no store
    _close_rotated(dq, q_want, lo.rms_norm_ref(q, w, EPS), "q")
    _close_rotated(dk, k_want, lo.rms_norm_ref(k, w, EPS), "k")


@pytest.mark.parametrize("N,hidden", [(1, 1024), (256, 4096), (33, 5120), (5, 8192), (1000, 1024), (3, 136)])
def test_add_rms_norm(gpu, N, hidden):
    from light_vllm_amd import ops
    gen = torch.Generator().manual_seed(hidden + N)
    x, r = rand_bf16(gen, N, hidden), rand_bf16(gen, N, hidden, scale=3.0)
    w = (1 + 0.2 * torch.randn(hidden, generator=gen)).to(torch.bfloat16)
    want = lo.rms_norm_ref(x, w, EPS)
    _close(ops.add_rms_norm(x.to(gpu), None, w.to(gpu), EPS), want, "rms")
    want, res_want = lo.add_rms_norm_ref(x, r, w, EPS)
    dr = r.to(gpu)
    dx = x.to(gpu)
    out = ops.add_rms_norm(dx, dr, w.to(gpu), EPS, out=dx)    # in place over x, as a caller may
    _close(out, want, "add_rms")
    assert _bits_equal(dr, res_want)


def test_add_rms_norm_strided_rows(gpu):
    from light_vllm_amd import ops
    gen = torch.Generator().manual_seed(3)
    big = rand_bf16(gen, 17, 3 * 1024)
    x = big[:, 1024:2048]
    w = torch.ones(1024, dtype=torch.bfloat16)
    got = ops.add_rms_norm(big.to(gpu)[:, 1024:2048], None, w.to(gpu), EPS)
    _close(got, lo.rms_norm_ref(x, w, EPS), "rms strided")


@pytest.mark.parametrize("N,inter", [(1, 3072), (256, 12288), (77, 25600 // 8), (4096, 3072), (2, 8)])
def test_silu_and_mul(gpu, N, inter):
    from light_vllm_amd import ops
    gen = torch.Generator().manual_seed(inter)
    x = rand_bf16(gen, N, 2 * inter, scale=2.5)
    got = ops.silu_and_mul(x.to(gpu))
    _close(got, lo.silu_and_mul_ref(x), "silu_and_mul")


def test_hidden_above_limit_is_refused(gpu):
    from light_vllm_amd import ops
    x = torch.zeros(2, 8200, dtype=torch.bfloat16, device=gpu)
    with pytest.raises(RuntimeError, match="hidden"):
        ops.add_rms_norm(x, None, torch.ones(8200, dtype=torch.bfloat16, device=gpu), EPS)

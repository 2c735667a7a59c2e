"""BASELINE config 4 at full size: Qwen3-8B under TP = 2 / 4 / 8 leaves each rank 16q/4kv, 8q/2kv or
4q/1kv heads, decoding against 32k-token contexts through a [B, 128]-column block table
(reference call: flash_attn_with_kvcache at lightvllm/layers/attention.py:172-175 with the static
[bs, ceil(max_model_len / 256)] table of engine/model_runner.py:507-514).

Per shape and batch size: the library's split heuristic and forced 1 / 32 / 256 splits; sampled rows
against the CPU oracle (max-abs <= 2e-2), every row through a size-independent property (permuting
the physical blocks together with the table must not change one output bit); then the same launch
captured in a hipGraph with padding rows and replayed on new buffer contents, bit-equal to eager."""
import pytest
import torch

from oracle.attention_oracle import decode_ref

pytestmark = pytest.mark.gpu
TOL = 2e-2
D, BLOCK, CTX, MAXBLK = 128, 256, 32768, 128
SHAPES = [(16, 4), (8, 2), (4, 1)]


def _device_case(gpu, B, Hq, Hkv, seed, lens=None):
    """Caches drawn on the device (a 64 x 32k x 4-head cache is 2 x 2.1 GB); every sequence owns
    128 physical blocks scattered by a seeded permutation."""
    g = torch.Generator(device=gpu).manual_seed(seed)
    nblk = B * MAXBLK + 3
    kc = torch.randn(nblk, BLOCK, Hkv, D, generator=g, device=gpu, dtype=torch.bfloat16)
    vc = torch.randn(nblk, BLOCK, Hkv, D, generator=g, device=gpu, dtype=torch.bfloat16)
    q = torch.randn(B, Hq, D, generator=g, device=gpu, dtype=torch.bfloat16)
    perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(seed))[:B * MAXBLK]
    bt = perm.view(B, MAXBLK).to(torch.int32).to(gpu)
    lens = [CTX] * B if lens is None else lens
    ctx = torch.tensor(lens, dtype=torch.int32, device=gpu)
    return dict(q=q, k_cache=kc, v_cache=vc, context_lens=ctx, block_table=bt, scale=D ** -0.5)


def _oracle_rows(d, rows):
    """decode_ref on a few rows: their blocks are gathered into a compact CPU cache."""
    out = []
    for b in rows:
        idx = d["block_table"][b].long()
        kc, vc = d["k_cache"][idx].cpu(), d["v_cache"][idx].cpu()
        bt = torch.arange(MAXBLK, dtype=torch.int32).view(1, -1)
        out.append(decode_ref(d["q"][b:b + 1].cpu(), kc, vc, d["context_lens"][b:b + 1].cpu(), bt, d["scale"]))
    return torch.cat(out)


def _run(ops, d, nsplit, **kw):
    return ops.decode_attention(d["q"], d["k_cache"], d["v_cache"], d["context_lens"], d["block_table"],
                                d["scale"], num_splits=nsplit, **kw)


@pytest.mark.parametrize("B", [1, 8, 64])
@pytest.mark.parametrize("Hq,Hkv", SHAPES)
def test_config4_decode_32k(gpu, Hq, Hkv, B):
    from light_vllm_amd import ops
    lens = [CTX] * B
    if B >= 8:      # ragged tail: the full table, one token short of it, a block boundary, a single token
        lens[1], lens[2], lens[3], lens[-1] = CTX - 1, 20000, 256 * 77, 1
    d = _device_case(gpu, B, Hq, Hkv, seed=400 + B + Hq, lens=lens)
    rows = sorted({0, 1 % B, 2 % B, B - 1})
    ref = _oracle_rows(d, rows)
    outs = {}
    for nsplit in (0, 1, 32, 256):
        outs[nsplit] = _run(ops, d, nsplit)
        torch.cuda.synchronize()
        err = (outs[nsplit][rows].float().cpu() - ref).abs().max().item()
        assert err <= TOL, f"splits {nsplit}: max-abs {err}"
    # both kernels serve these group sizes (4): the vector-ALU one must agree with the oracle too
    out_valu = _run(ops, d, 0, kernel="valu")
    torch.cuda.synchronize()
    assert (out_valu[rows].float().cpu() - ref).abs().max().item() <= TOL
    # all rows: a permutation of the physical blocks (table permuted with them) changes no bit
    nblk = d["k_cache"].size(0)
    perm = torch.randperm(nblk, generator=torch.Generator().manual_seed(9)).to(gpu)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nblk, device=gpu)
    d2 = dict(d, k_cache=d["k_cache"][perm], v_cache=d["v_cache"][perm],
              block_table=inv[d["block_table"].long()].to(torch.int32))
    out2 = _run(ops, d2, 0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), out2.view(torch.int16))
    # all rows: every split count describes the same function
    for nsplit in (1, 32, 256):
        assert (outs[nsplit].float() - outs[0].float()).abs().max().item() <= TOL


@pytest.mark.parametrize("Hq,Hkv", SHAPES)
def test_config4_hipgraph_with_padding_rows(gpu, Hq, Hkv):
    """The decode launch as ModelRunner captures it: static [8, 128] table, 5 live rows and 3
    padding rows (context length 0, table row zeros, their own index in the launch order), replayed
    after the buffers' CONTENTS changed. Each replay must equal the eager call on the same data bit
    for bit, and the padding rows must come out zero."""
    from light_vllm_amd import ops
    bucket, live = 8, 5
    d = _device_case(gpu, bucket, Hq, Hkv, seed=500 + Hq)
    steps = [[CTX, 30001, 1, 255, 4096], [CTX - 1, 30002, 2, 256, 4097]]
    ctx_buf = torch.zeros(bucket, dtype=torch.int32, device=gpu)
    bt_buf = torch.zeros(bucket, MAXBLK, dtype=torch.int32, device=gpu)
    q_buf = torch.zeros(bucket, Hq, D, dtype=torch.bfloat16, device=gpu)
    order_buf = torch.arange(bucket, dtype=torch.int32, device=gpu)
    out_buf = torch.empty_like(q_buf)
    nsplit = ops.decode_num_splits(bucket, Hkv, MAXBLK, BLOCK)
    ops.reserve_decode_workspace(gpu, ops.decode_workspace_bytes(bucket, Hq, D, nsplit))

    def fill(lens, with_order):
        ctx_buf.zero_()
        bt_buf.zero_()
        ctx_buf[:live] = torch.tensor(lens, dtype=torch.int32, device=gpu)
        bt_buf[:live] = d["block_table"][:live]
        q_buf.copy_(d["q"])
        if with_order:      # longest first, padding rows behind the live ones
            order = sorted(range(live), key=lambda j: -lens[j]) + list(range(live, bucket))
            order_buf.copy_(torch.tensor(order, dtype=torch.int32, device=gpu))
        else:               # a negative first entry: the kernel's default order
            order_buf[0] = -1

    def launch():
        ops.decode_attention(q_buf, d["k_cache"], d["v_cache"], ctx_buf, bt_buf, d["scale"], out=out_buf,
                             seq_order=order_buf)

    fill(steps[0], True)
    launch()                                 # warm run outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        launch()
    for lens, with_order in ((steps[0], True), (steps[1], False), (steps[1], True)):
        fill(lens, with_order)
        out_buf.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        got = out_buf.clone()
        eager = ops.decode_attention(q_buf, d["k_cache"], d["v_cache"], ctx_buf, bt_buf, d["scale"],
                                     seq_order=order_buf if with_order else None)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), eager.view(torch.int16))
        assert got[live:].abs().sum().item() == 0
        dd = dict(d, context_lens=ctx_buf, block_table=bt_buf)
        ref = _oracle_rows(dd, [0, 2, 4])
        assert (got[[0, 2, 4]].float().cpu() - ref).abs().max().item() <= TOL

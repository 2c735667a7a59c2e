"""Writes tests/golden/layers_small.safetensors: seeded inputs and the outputs of the REFERENCE's
own RMSNorm (rms_forward / add_rms_forward) and RotaryEmbedding modules, run eagerly on CPU from
/root/reference. Run in the build container only (TEST INFRASTRUCTURE); the GPU box reads only the
committed fixture.

    TORCHDYNAMO_DISABLE=1 python oracle/gen_layers_golden.py
"""
from __future__ import annotations

import os
import sys

os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REFERENCE)

import torch  # noqa: E402
from safetensors.torch import save_file  # noqa: E402


def inputs():
    g = torch.Generator().manual_seed(77)
    bf = torch.bfloat16
    N, Hq, Hkv, D, hidden = 6, 4, 2, 128, 1024
    t = {
        "q": (torch.randn(N, Hq, D, generator=g) * 3).to(bf),
        "k": (torch.randn(N, Hkv, D, generator=g) * 0.5).to(bf),
        "q_weight": (1 + 0.2 * torch.randn(D, generator=g)).to(bf),
        "k_weight": (1 + 0.2 * torch.randn(D, generator=g)).to(bf),
        "positions": torch.tensor([0, 1, 255, 256, 4095, 40959], dtype=torch.int64),
        "x": torch.randn(N, hidden, generator=g).to(bf),
        "residual": (torch.randn(N, hidden, generator=g) * 2).to(bf),
        "weight": (1 + 0.1 * torch.randn(hidden, generator=g)).to(bf),
    }
    return t


def main():
    from lightvllm.layers.layernorm import RMSNorm
    from lightvllm.layers.rotary_embedding import RotaryEmbedding
    eps, D, max_pos, base = 1e-6, 128, 40960, 1000000.0
    t = inputs()
    torch.set_default_dtype(torch.bfloat16)
    qn, kn, hn = RMSNorm(D, eps), RMSNorm(D, eps), RMSNorm(t["x"].size(-1), eps)
    torch.set_default_dtype(torch.float32)
    qn.weight.data.copy_(t["q_weight"])
    kn.weight.data.copy_(t["k_weight"])
    hn.weight.data.copy_(t["weight"])
    rope = RotaryEmbedding(D, D, max_pos, base)
    with torch.no_grad():
        q_normed = qn.rms_forward(t["q"].clone())
        k_normed = kn.rms_forward(t["k"].clone())
        N = t["q"].size(0)
        q_rot, k_rot = rope(t["positions"], q_normed.view(N, -1).clone(), k_normed.view(N, -1).clone())
        rms_out = hn.rms_forward(t["x"].clone())
        add_out, add_res = hn.add_rms_forward(t["x"].clone(), t["residual"].clone())
    out = dict(t)
    out.update(q_normed=q_normed, k_normed=k_normed, q_rot=q_rot.view_as(t["q"]), k_rot=k_rot.view_as(t["k"]),
               rms_out=rms_out, add_out=add_out, add_res=add_res,
               cos_sin_rows=rope.cos_sin_cache[t["positions"]].reshape(N, -1).float())
    out = {k: v.contiguous() for k, v in out.items()}
    path = os.path.join(ROOT, "tests", "golden", "layers_small.safetensors")
    save_file(out, path, metadata={"eps": str(eps), "max_position": str(max_pos), "base": str(base)})
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()

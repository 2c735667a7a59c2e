import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from light_vllm_amd import ops
from oracle.attention_oracle import prefill_ref
torch.manual_seed(0)
dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Hq = Hkv = 1
q = torch.randn(L, Hq, 128).to(torch.bfloat16); k = torch.randn(L, Hkv, 128).to(torch.bfloat16); v = torch.randn(L, Hkv, 128).to(torch.bfloat16)
cu = torch.tensor([0, L], dtype=torch.int32)
ref = prefill_ref(q, k, v, cu, cu, 128 ** -0.5, None)
out = ops.prefill_attention(q.to(dev), k.to(dev), v.to(dev), cu.to(dev), cu.to(dev), L, L, 128 ** -0.5).float().cpu()
err = (out - ref).abs()[:, 0, :]
print("max err", err.max().item())
rows = err.max(dim=1).values
print("rows with err>0.02:", (rows > 0.02).nonzero().flatten().tolist()[:80])
cols = err.max(dim=0).values
print("cols with err>0.02:", (cols > 0.02).nonzero().flatten().tolist())
r = int(rows.argmax()); print("row", r, "out", out[r, 0, :8].tolist(), "ref", ref[r, 0, :8].tolist())
print("ratio row r:", (out[r,0,:8]/ref[r,0,:8]).tolist())

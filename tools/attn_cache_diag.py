"""Diagnostic: packed-row vs cache-layout K/V through attn_fwd — which rows / heads differ, and is a call deterministic."""
import sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
B.load_library()
DEV = "cuda"
D, H, causal = 128, 3, True
lens = [70, 1, 129, 200]
M, max_len = sum(lens), 256
cu = [0]
for n in lens:
    cu.append(cu[-1] + n)
g = torch.Generator(device="cpu").manual_seed(7)
q, k, v = (torch.randn(M, H * D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(3))
cu_t = torch.tensor(cu, dtype=torch.int32, device=DEV)
outs = []
for rep in range(3):
    ref = torch.empty(M, H * D, dtype=torch.bfloat16, device=DEV)
    B.attn_fwd(q, k, v, ref, cu_t, max(lens), H, D, D ** -0.5, causal=causal)
    outs.append(ref.clone())
print("packed deterministic:", torch.equal(outs[0], outs[1]), torch.equal(outs[0], outs[2]))
for fill in (float("nan"), 0.0):
    kc = torch.full((len(lens), H, max_len, D), fill, dtype=torch.bfloat16, device=DEV)
    vc = torch.full_like(kc, fill)
    for s, n in enumerate(lens):
        kc[s, :, :n] = k[cu[s]:cu[s + 1]].view(n, H, D).transpose(0, 1)
        vc[s, :, :n] = v[cu[s]:cu[s + 1]].view(n, H, D).transpose(0, 1)
    out = torch.empty_like(outs[0])
    B.attn_fwd(q, kc, vc, out, cu_t, max(lens), H, D, D ** -0.5, causal=causal, kv_cache_max_len=max_len)
    bad = (out != outs[0]) | (out.isnan())
    rows = bad.any(1).nonzero().flatten().tolist()
    print("fill", fill, "rows differing:", len(rows), rows[:40], "nan:", int(out.isnan().sum()))
    if rows:
        r = rows[0]
        print(" row", r, "heads:", bad[r].view(H, D).any(1).tolist(), "maxdiff", float((out[r].float() - outs[0][r].float()).abs().max()))

"""How far is attn_fwd from an f32-softmax reference on the same bf16 q/k/v (before / after the final bf16 rounding of O)?"""
import sys, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
B.load_library()
torch.manual_seed(0)
for (D, H, L, causal) in [(128, 2, 155, True), (128, 2, 376, True), (64, 2, 1500, False)]:
    qkv = (torch.randn(L, 3 * H * D, device="cuda") * 1.0).to(torch.bfloat16)
    q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    out = torch.empty(L, H * D, dtype=torch.bfloat16, device="cuda")
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    B.attn_fwd(q, k, v, out, cu, L, H, D, D ** -0.5, causal=causal)
    qs, ks, vs = (t.double().view(L, H, D).transpose(0, 1) for t in (q, k, v))
    sc = qs @ ks.transpose(1, 2) * D ** -0.5
    if causal:
        i = torch.arange(L, device="cuda")
        sc = sc.masked_fill((i[None, :] > i[:, None])[None], float("-inf"))
    ref = (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(L, H * D)
    refb = ref.to(torch.bfloat16).double()
    o = out.double()
    print(f"D={D} L={L} causal={causal}: rel-L2 vs exact {float((o-ref).norm()/ref.norm()):.2e}; "
          f"exact rounded to bf16 vs exact {float((refb-ref).norm()/ref.norm()):.2e}; kernel vs rounded-exact {float((o-refb).norm()/ref.norm()):.2e}; "
          f"elements differing from rounded-exact {float((o!=refb).double().mean()):.3f}")

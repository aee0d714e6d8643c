"""rope_kv at the prefill shape of the bench (48128 rows, 32 heads x 128) and at decode (128 rows)."""
import sys, os, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
DEV = "cuda"
B.load_library()
H, D, max_len, nseq, S = 32, 128, 448, 128, 376
for M in (nseq * S, 128):
    qkv = torch.randn(M, 3 * H * D, device=DEV).to(torch.bfloat16)
    pos = (torch.arange(M, device=DEV) % S).to(torch.int32)
    sid = (torch.arange(M, device=DEV) // S % nseq).to(torch.int32)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=DEV).float() / D))
    ang = torch.arange(2048, device=DEV).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    kc = torch.zeros(nseq, H, max_len, D, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    f = lambda: B.rope_kv(qkv, H * D, 2 * H * D, cos, sin, pos, sid, kc, vc, H, D, max_len)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"rope_kv M={M}: {t*1e6:.1f} us  {M*H*D*2*7/t/1e12:.2f} TB/s (q,k read+write, v read, k,v cache write)")

"""Runs the prefill attention kernel alone at the bench shapes (for rocprofv3 --pmc): whisper / beats / llama."""
import sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B

import os
DEV = "cuda"
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]      # A/B against another build of the library (same box, same call)
B.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "whisper"
nseq, L, H, D, causal = {"whisper": (64, 1500, 20, 64, False), "llama": (64, 376, 32, 128, True), "qwen": (32, 1264, 32, 128, True), "llama600": (64, 600, 40, 128, True), "beats": (64, 1496, 12, 64, False),
                         "beats_bias": (64, 1496, 12, 64, False)}[which]
nseq = int(os.environ.get("ICL_ATTN_NSEQ", nseq))
total = nseq * L
qkv = torch.randn(total, 3 * H * D, device=DEV).to(torch.bfloat16)
out = torch.empty(total, H * D, dtype=torch.bfloat16, device=DEV)
cu = torch.arange(0, total + 1, L, dtype=torch.int32, device=DEV)
q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
kw = {}
if which == "beats_bias":       # gated relative-position bias (BEATs): table [H, 2*span-1], gate [rows, H]
    kw = dict(rel_bias=torch.randn(H, 2 * L - 1, device=DEV), rel_gate=torch.rand(total, H, device=DEV) * 2, rel_span=L)
for _ in range(3):
    B.attn_fwd(q, k, v, out, cu, L, H, D, D ** -0.5, causal=causal, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    B.attn_fwd(q, k, v, out, cu, L, H, D, D ** -0.5, causal=causal, **kw)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
flops = 4 * nseq * H * L * L * D * (0.5 if causal else 1.0)
print(f"attn {which}: {t*1e3:.3f} ms {flops/t/1e12:.1f} TF/s")

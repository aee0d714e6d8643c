"""QKV projection: icl_gemm_bf16 + icl_rope_kv_bf16 vs the fused icl_gemm_rope_kv_bf16 at the bench's prefill shape."""
import importlib
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
B = importlib.import_module("icl-speech-text-llm_amd.runtime.binding")
DEV = "cuda:0"


def main():
    nseq, T, H, D, K = (int(a) for a in (sys.argv[1:6] + [128, 376, 32, 128, 4160][len(sys.argv) - 1:]))
    M, hd, max_len = nseq * T, H * D, T + 10
    x = (torch.randn(M, K, device=DEV) * 0.5).to(torch.bfloat16)
    w = (torch.randn(3 * hd, K, device=DEV) * 0.02).to(torch.bfloat16)
    pos = torch.arange(T, dtype=torch.int32, device=DEV).repeat(nseq)
    sid = torch.arange(nseq, dtype=torch.int32, device=DEV).repeat_interleave(T)
    inv = 1.0 / (10000 ** (torch.arange(0, D, 2, device=DEV).float() / D))
    ang = torch.arange(max_len, device=DEV).float()[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    kc = torch.zeros(nseq, H, max_len, D, dtype=torch.bfloat16, device=DEV)
    vc = torch.zeros_like(kc)
    out = torch.empty(M, 3 * hd, dtype=torch.bfloat16, device=DEV)
    rope = (hd, 2 * hd, cos, sin, pos, sid, kc, vc, H, D, max_len)

    def unfused():
        B.gemm(x, w, out, tile=3)
        B.rope_kv(out, hd, 2 * hd, cos, sin, pos, sid, kc, vc, H, D, max_len)

    def fused():
        B.gemm(x, w, out, tile=3, rope=rope)

    def fused_nocache():
        B.gemm(x, w, out, tile=3, rope=(hd, 2 * hd, cos, sin, pos, None, None, None, H, D, max_len))

    for name, fn in (("gemm + rope_kv", unfused), ("fused", fused), ("fused, no cache", fused_nocache)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:18s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()

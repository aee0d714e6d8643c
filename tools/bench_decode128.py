"""Decode GEMMs at 64 < M <= 128 with ROTATING weights (> 256 MiB per shape): 64x64 tile + split-K (the previous choice) vs the
M<=128 decode tile (tile 5) over a split-K sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
from bench_decode import time_rot

DEV = "cuda"
for M in (128, 96):
    total = {}
    for name, N, K, sk2 in [("qkv", 12288, 4160, 2), ("o", 4096, 4096, 4), ("gu", 22016, 4096, 1), ("down", 4096, 11008, 4)]:
        nrot = max(8, int(600e6 / (N * K * 2)) + 1)
        ws_ = [(torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nrot)]
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
        wp_ = [B.pack_decode_weights(w) for w in ws_]
        res = []
        t = time_rot([(lambda w=w: B.gemm(a, w, out, tile=2, split_k=sk2, workspace=ws)) for w in ws_])
        res.append(f"64x64/sk{sk2}: {t*1e6:5.1f}us {N*K*2/t/1e12:4.2f}")
        total["old"] = total.get("old", 0) + t
        best = 1e9
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > K // 64:
                continue
            t = time_rot([(lambda w=w: B.gemm(a, w, out, tile=5, split_k=sk, workspace=ws, N=N)) for w in wp_])
            res.append(f"sk{sk}: {t*1e6:5.1f} {N*K*2/t/1e12:4.2f}")
            best = min(best, t)
        total["new"] = total.get("new", 0) + best
        print(f"M={M:3d} {name:4s} N={N:5d} K={K:5d} | " + " | ".join(res), flush=True)
        del ws_, wp_
    print(f"M={M}: per layer {total['old']*1e6:.1f} us -> {total['new']*1e6:.1f} us (best split per GEMM)", flush=True)

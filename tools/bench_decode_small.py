"""Decode GEMMs at M <= 64 with ROTATING weights: skinny on row-major W (tile 4) / on the decode-packed copy (tile 6), the 64x64
tile + split-K (tile 2), and the decode tile with 64-row blocks (tile 5) over a split sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
from bench_decode import time_rot

DEV = "cuda"
shapes = [("qkv", 12288, 4160), ("o", 4096, 4096), ("gu", 22016, 4096), ("down", 4096, 11008)]
bufs = {}
for name, N, K in shapes:
    nrot = max(4, int(300e6 / (N * K * 2)) + 1)
    ws_ = [(torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nrot)]
    bufs[name] = (ws_, [B.pack_decode_weights(w) for w in ws_])
for M in (1, 4, 8, 16, 32, 64):
    tot = {}
    for name, N, K in shapes:
        ws_, wp_ = bufs[name]
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
        r = {}
        r["t4"] = time_rot([(lambda w=w: B.gemm(a, w, out, tile=4)) for w in ws_])
        r["t6"] = time_rot([(lambda w=w: B.gemm(a, w, out, tile=6, N=N)) for w in wp_])
        tiles = ((N + 63) // 64)
        sk2 = max(1, min(K // 512, (512 + tiles - 1) // tiles, 16))
        r["t2"] = time_rot([(lambda w=w: B.gemm(a, w, out, tile=2, split_k=sk2, workspace=ws)) for w in ws_])
        best5 = None
        for sk in (1, 2, 3, 4, 6, 8):
            t = time_rot([(lambda w=w: B.gemm(a, w, out, tile=5, split_k=sk, workspace=ws, N=N)) for w in wp_])
            if best5 is None or t < best5[0]:
                best5 = (t, sk)
        r["t5"] = best5[0]
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
        print(f"M={M:2d} {name:4s} | skinny {r['t4']*1e6:6.1f} | skinny packed {r['t6']*1e6:6.1f} | 64x64 sk{sk2} {r['t2']*1e6:6.1f} | "
              f"decode tile sk{best5[1]} {r['t5']*1e6:6.1f} us", flush=True)
    print(f"M={M:2d} per layer: " + "  ".join(f"{k} {v*1e6:6.1f}" for k, v in tot.items()) + "  us", flush=True)

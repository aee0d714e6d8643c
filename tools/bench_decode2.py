import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
from bench_decode import time_rot
DEV = "cuda"
for M in (96, 128, 256):
    for N, K in [(4096, 4096), (12288, 4160), (22016, 4096), (4096, 11008)]:
        nrot = max(8, int(600e6 / (N * K * 2)) + 1)
        ws_ = [(torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nrot)]
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(32 * M * N, dtype=torch.float32, device=DEV)
        res = []
        for tile in (1, 2):
            for sk in (1, 4, 8, 16):
                if sk > K // 64: continue
                t = time_rot([(lambda w=w: B.gemm(a, w, out, tile=tile, split_k=sk, workspace=ws)) for w in ws_])
                res.append(f"t{tile}/sk{sk}: {t*1e6:5.1f}us {N*K*2/t/1e12:4.2f}")
        print(f"M={M:3d} N={N:5d} K={K:5d} | " + " | ".join(res), flush=True)
        del ws_

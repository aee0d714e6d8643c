"""BEATs grouped positional conv as a batched GEMM (M = 1496 per clip, N = 48, K = 6144, batch = 128): tile 1 / 2 / 3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B

DEV = "cuda"
n, T, d, cpg = 128, 1496, 768, 48
xg = (torch.randn((n * (T + 128)) * d, device=DEV) * 0.5).to(torch.bfloat16)
w = (torch.randn(cpg, 128 * cpg, device=DEV) * 0.02).to(torch.bfloat16)
x = torch.randn(n * T, d, device=DEV)
y = torch.empty(n * T, d, device=DEV)
bias = torch.randn(cpg, device=DEV)
for tile in (1, 2, 3):
    def run():
        B.gemm(xg, w, y[:, :cpg], bias=bias, gelu=True, residual=x[:, :cpg], M=T, K=128 * cpg, lda=cpg, batch=n,
               stride_a=(T + 128) * d, stride_c=T * d, stride_r=T * d, tile=tile)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"tile {tile}: {t*1e6:7.1f} us  {2*n*T*cpg*128*cpg/t/1e12:6.1f} TF/s useful", flush=True)

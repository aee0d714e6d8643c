"""GEMM microbench at the default bench's shapes (micro-batch 128): tile 1 / tile 3 / hipBLASLt (torch.matmul), with the
epilogues the path uses.  Random operands (DVFS-honest)."""
import sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B

import os
DEV = "cuda"
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


shapes = [(48128, 12288, 4160, "llama qkv", 0), (48128, 4096, 4096, "llama o +res", 4), (48128, 22016, 4096, "llama gate/up swiglu", 8),
          (48128, 4096, 11008, "llama down +res", 4), (192000, 3840, 1280, "whisper qkv +bias", 1), (192000, 1280, 1280, "whisper o +bias+res", 5),
          (192000, 5120, 1280, "whisper fc1 +bias+gelu", 3), (192000, 1280, 5120, "whisper fc2 +bias+res", 5),
          (191488, 2304, 768, "beats qkv +bias", 1), (191488, 768, 768, "beats o", 1), (191488, 3072, 768, "beats fc1 +gelu", 3),
          (191488, 768, 3072, "beats fc2", 1), (8192, 8192, 8192, "8192^3", 0)]
B.load_library()
for M, N, K, tag, epi in shapes:
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    n_out = N // 2 if epi & 8 else N
    out = torch.empty(M, n_out, dtype=torch.float32 if epi & 4 else torch.bfloat16, device=DEV)
    bias = torch.randn(N, device=DEV) if epi & 1 else None
    res = torch.randn(M, n_out, device=DEV) if epi & 4 else None
    kw = dict(bias=bias, residual=res, gelu=bool(epi & 2), swiglu=bool(epi & 8))
    line = f"{tag:24s} M={M:6d} N={N:5d} K={K:5d}:"
    for tile in (1, 3):
        try:
            t = timeit(lambda: B.gemm(a, w, out, tile=tile, **kw))
            line += f"  tile{tile} {t*1e3:7.3f} ms {2*M*N*K/t/1e12:7.1f} TF/s"
        except Exception as e:
            line += f"  tile{tile} ERR {str(e)[:40]}"
    out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: B.gemm(a, w, out2, tile=3))
    line += f" | plain tile3 {2*M*N*K/t/1e12:7.1f}"
    t = timeit(lambda: torch.matmul(a, w.t()))
    line += f" | hipblaslt {2*M*N*K/t/1e12:7.1f} TF/s"
    print(line, flush=True)
    del a, w, out, out2, bias, res

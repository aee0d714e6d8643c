import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
from bench_kernels import timeit
DEV = "cuda"
B.load_library()
for (M, N, K) in [(48000, 5120, 1280), (48000, 3840, 1280), (48000, 1280, 1280), (48000, 1280, 5120), (12032, 4096, 4096)]:
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    out32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    res = torch.randn(M, N, device=DEV)
    for tile in (1, 3):
        r = {}
        r["plain->bf16"] = timeit(lambda: B.gemm(a, w, out16, tile=tile))
        r["bias->bf16"] = timeit(lambda: B.gemm(a, w, out16, bias=bias, tile=tile))
        r["bias+gelu->bf16"] = timeit(lambda: B.gemm(a, w, out16, bias=bias, gelu=True, tile=tile))
        r["plain->f32"] = timeit(lambda: B.gemm(a, w, out32, tile=tile))
        r["bias+res(f32,inplace)->f32"] = timeit(lambda: B.gemm(a, w, res, bias=bias, residual=res, tile=tile))
        print(f"M={M} N={N} K={K} tile={tile}: " + "  ".join(f"{k}: {2*M*N*K/t/1e12:6.0f} TF" for k, t in r.items()), flush=True)

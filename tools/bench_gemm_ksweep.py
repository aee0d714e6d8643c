"""Per-tile fixed cost of the 256x256 GEMM: K sweep at the Whisper qkv shape (plain bf16 output)."""
import sys, os, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
DEV = "cuda"
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
B.load_library()

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

M, N = 192000, 3840
tiles_per_cu = (M // 256) * (N // 256) / 256
for K in (128, 256, 512, 1024, 1280, 2048, 4096):
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: B.gemm(a, w, out, tile=3))
    print(f"K={K:5d}: {t*1e3:7.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s  per tile-slot {t/tiles_per_cu*1e6:6.2f} us", flush=True)
    del a, w, out

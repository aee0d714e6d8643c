"""Norm kernels at the bench's shapes (micro-batch 128): in-path LN / RMS calls with rotating inputs (HBM rates)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
DEV = "cuda"
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
B.load_library()

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

for (M, N, tag) in [(192000, 1280, "whisper LN"), (191488, 768, "beats LN"), (48128, 4096, "llama rms"), (30720, 5120, "13B rms")]:
    x = torch.randn(M, N, device=DEV)
    g, b = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    out = torch.empty(M, N + 64, dtype=torch.bfloat16, device=DEV)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    t = timeit(lambda: B.layernorm(x, g, b, out, 1e-5, N=N))
    t2 = timeit(lambda: B.rmsnorm(x, g, out, 1e-5, N=N))
    t3 = timeit(lambda: B.layernorm(x, g, b, o32, 1e-5, res=x, alpha=1.5, out2=out, N=N))
    print(f"{tag:12s} [{M},{N}] LN {t*1e6:7.1f} us {M*N*6/t/1e12:5.2f} TB/s | RMS {t2*1e6:7.1f} us {M*N*6/t2/1e12:5.2f} | LN+res dual {t3*1e6:7.1f} us {M*N*14/t3/1e12:5.2f}", flush=True)

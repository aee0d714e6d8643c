import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
from bench_kernels import timeit
DEV = "cuda"
B.load_library()
for (M, N, tag) in [(48000, 1280, "whisper LN"), (47872, 768, "beats LN"), (12032, 4096, "llama rms"), (32, 4096, "llama rms decode")]:
    x = torch.randn(M, N, device=DEV)
    g, b = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    out = torch.empty(M, N + 64, dtype=torch.bfloat16, device=DEV)
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    t = timeit(lambda: B.layernorm(x, g, b, out, 1e-5, N=N), iters=50)
    print(f"{tag:18s} LN  f32->bf16 [{M},{N}]: {t*1e6:8.1f} us  {M*N*6/t/1e12:5.2f} TB/s")
    t = timeit(lambda: B.rmsnorm(x, g, out, 1e-5, N=N), iters=50)
    print(f"{tag:18s} RMS f32->bf16 [{M},{N}]: {t*1e6:8.1f} us  {M*N*6/t/1e12:5.2f} TB/s")
    t = timeit(lambda: B.layernorm(x, g, b, o32, 1e-5, res=x, alpha=1.5, out2=out, N=N), iters=50)
    print(f"{tag:18s} LN+res f32->f32+bf16: {t*1e6:8.1f} us  {M*N*14/t/1e12:5.2f} TB/s")

"""Decode-shape GEMM benchmark with ROTATING weights (8 distinct matrices, > 256 MiB in total) so the Infinity Cache
cannot serve the stream: the numbers are HBM rates, like in the real decode loop."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B
DEV = "cuda"
B.load_library()

def time_rot(fns, iters=5):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * len(fns)) * 1e-3

for M in (1, 8, 32, 64, 128):
    for N, K, sk in [(4096, 4096, 8), (12288, 4160, 4), (22016, 4096, 2), (4096, 11008, 8)]:
        nrot = max(8, int(600e6 / (N * K * 2)) + 1)
        ws_ = [(torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(nrot)]
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ws = torch.empty(16 * M * N, dtype=torch.float32, device=DEV)
        res = []
        for label, kw in (("64x64 sk%d" % sk, dict(tile=2, split_k=sk, workspace=ws)), ("64x64 sk%d" % (2 * sk), dict(tile=2, split_k=2 * sk, workspace=ws)),
                          ("skinny", dict(tile=4))):
            if kw.get("tile") == 4 and M > 64:
                continue
            t = time_rot([(lambda w=w, kw=kw: B.gemm(a, w, out, **kw)) for w in ws_])
            res.append(f"{label}: {t*1e6:6.1f} us {N*K*2/t/1e12:4.2f} TB/s")
        print(f"M={M:3d} N={N:5d} K={K:5d} | " + " | ".join(res), flush=True)
        del ws_

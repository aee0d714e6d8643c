"""Micro-benchmarks of the hot kernels (GPU box only): python tools/bench_kernels.py [gemm|attn|all]"""
import sys
import os
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icl_speech_text_llm_amd.runtime.binding as B

DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
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


def bench_gemm():
    shapes = [  # (M, N, K, tag)
        (6016, 12288, 4096, "llama qkv  B=16"), (6016, 4096, 4096, "llama o    B=16"),
        (6016, 22016, 4096, "llama gate/up B=16"), (6016, 4096, 11008, "llama down B=16"),
        (376, 4096, 4096, "llama o    B=1"), (24000, 3840, 1280, "whisper qkv B=16"),
        (24000, 5120, 1280, "whisper fc1 B=16"), (24000, 1280, 5120, "whisper fc2 B=16"),
        (12032, 12288, 4160, "llama qkv  B=32"), (12032, 4096, 4096, "llama o    B=32"),
        (12032, 22016, 4096, "llama gate/up B=32"), (12032, 4096, 11008, "llama down B=32"),
        (48000, 3840, 1280, "whisper qkv B=32"), (48000, 1280, 1280, "whisper o  B=32"),
        (48000, 5120, 1280, "whisper fc1 B=32"), (48000, 1280, 5120, "whisper fc2 B=32"),
        (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3"),
    ]
    for M, N, K, tag in shapes:
        a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        for tile in (1, 3):
            t = timeit(lambda: B.gemm(a, w, out, tile=tile))
            print(f"gemm tile={tile} {tag:22s} M={M:6d} N={N:6d} K={K:6d}  {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
        t = timeit(lambda: torch.matmul(a, w.t()))
        print(f"gemm hipblaslt(torch)   {tag:22s}                              {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:7.1f} TF/s", flush=True)
    # decode shapes
    for M in (1, 32, 64):
        for N, K, sk in [(4096, 4096, 8), (12288, 4160, 4), (22016, 4096, 2), (4096, 11008, 8), (32001, 4096, 1)]:
            a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
            w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
            out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
            ws = torch.empty(sk * M * N, dtype=torch.float32, device=DEV)
            t2 = timeit(lambda: B.gemm(a, w, out, tile=2, split_k=sk, workspace=ws))
            t4 = timeit(lambda: B.gemm(a, w, out, tile=4))
            print(f"decode gemm M={M:2d} N={N:5d} K={K:5d}: 64x64+splitK{sk} {t2*1e6:7.1f} us {N*K*2/t2/1e12:5.2f} TB/s | skinny {t4*1e6:7.1f} us {N*K*2/t4/1e12:5.2f} TB/s", flush=True)


def bench_attn():
    for (nseq, L, H, D, causal, tag) in [(16, 1500, 20, 64, False, "whisper"), (16, 376, 32, 128, True, "llama prefill"),
                                         (16, 1496, 12, 64, False, "beats")]:
        total = nseq * L
        qkv = torch.randn(total, 3 * H * D, device=DEV).to(torch.bfloat16)
        out = torch.empty(total, H * D, dtype=torch.bfloat16, device=DEV)
        cu = torch.arange(0, total + 1, L, dtype=torch.int32, device=DEV)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
        t = timeit(lambda: B.attn_fwd(q, k, v, out, cu, L, H, D, D ** -0.5, causal=causal))
        flops = 4 * nseq * H * L * L * D * (0.5 if causal else 1.0)
        print(f"attn {tag:14s} nseq={nseq} L={L} H={H} D={D}: {t*1e3:8.3f} ms  {flops/t/1e12:7.1f} TF/s", flush=True)
    # decode attention
    Bn, H, D, max_len = 32, 32, 128, 400
    lens = torch.full((Bn,), 386, dtype=torch.int32, device=DEV)
    q = torch.randn(Bn, H * D, device=DEV).to(torch.bfloat16)
    kc = torch.randn(Bn, H, max_len, D, device=DEV).to(torch.bfloat16)
    vc = torch.randn(Bn, H, max_len, D, device=DEV).to(torch.bfloat16)
    out = torch.empty(Bn, H * D, dtype=torch.bfloat16, device=DEV)
    t = timeit(lambda: B.attn_decode(q, kc, vc, out, lens, H, D, max_len, D ** -0.5))
    print(f"attn decode B={Bn} len=386: {t*1e6:8.1f} us  {2*Bn*H*386*D*2/t/1e12:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    B.load_library()
    if what in ("gemm", "all"):
        bench_gemm()
    if what == "decode":
        import types
        src = open(__file__).read()
        bench_gemm.__globals__["__decode_only__"] = True
    if what in ("attn", "all"):
        bench_attn()

"""Times the decode tile (icl_gemm_bf16 tile 5) at the four per-layer decode shapes of Llama-2-7B for 128 and 256 rows, with
rotating weight copies so nothing is served from cache (per-layer microseconds and the weight stream rate)."""
import os, sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
B.load_library()
DEV, NCU, COPIES = "cuda", 256, 6
def sk5(N, K): return max(1, min(NCU // ((N + 127) // 128), K // 512))
shapes = [("qkv", 12288, 4160, False), ("o", 4096, 4096, False), ("gate/up", 22016, 4096, True), ("down", 4096, 11008, False)]
SWEEP = os.environ.get("ICL_SPLIT_SWEEP")          # e.g. "o:2,4,8,16;down:2,4,8,16;qkv:1,2,4" -> time each split at M = 256
if SWEEP:
    want = {kv.split(":")[0]: [int(x) for x in kv.split(":")[1].split(",")] for kv in SWEEP.split(";")}
    M = 256
    for name, N, K, sw in shapes:
        for split in want.get(name, []):
            g = torch.Generator().manual_seed(1)
            a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
            ws_ = [B.pack_decode_weights((torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)) for _ in range(COPIES)]
            wsk = torch.empty(split * M * N, device=DEV) if split > 1 else None
            out = torch.empty(M, N // 2 if sw else N, dtype=torch.bfloat16, device=DEV)
            def run(i): B.gemm(a, ws_[i % COPIES], out, swiglu=sw, tile=5, split_k=split, workspace=wsk, M=M, N=N)
            for i in range(COPIES): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10 * COPIES): run(i)
            e1.record(); torch.cuda.synchronize()
            print(f"M={M} {name:8s} split={split:2d}: {e0.elapsed_time(e1) / (10 * COPIES) * 1e3:7.1f} us")
    sys.exit(0)
T256 = os.environ.get("ICL_T256_SWEEP")            # e.g. "qkv:4,5,6;o:8,16;gate/up:2,3;down:8,16": the 256x256 tile with split-K at M = 256
if T256:                                           # (row-major weights), GEMM + its slab reduction, next to the decode tile's current choice
    want = {kv.split(":")[0]: [int(x) for x in kv.split(":")[1].split(",")] for kv in T256.split(";")}
    M = int(os.environ.get("ICL_T256_M", "256"))
    for name, N, K, sw in shapes:
        g = torch.Generator().manual_seed(1)
        a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        raw = [(torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV) for _ in range(COPIES)]
        packed = [B.pack_decode_weights(w) for w in raw]
        f32_out = name in ("o", "down")            # these two end in the f32 residual stream (+ the fused RMSNorm in the model)
        out = torch.empty(M, N // 2 if sw else N, dtype=torch.float32 if f32_out else torch.bfloat16, device=DEV)
        res = torch.randn(M, N, device=DEV) if f32_out else None
        for tile, splits in ((5, [sk5(N, K)]), (3, want.get(name, []))):
            for split in splits:
                wsk = torch.empty(split * M * N, device=DEV) if split > 1 else None
                ws_ = packed if tile == 5 else raw
                def run(i): B.gemm(a, ws_[i % COPIES], out, swiglu=sw, residual=res, tile=tile, split_k=split, workspace=wsk, M=M, N=N)
                for i in range(COPIES): run(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(10 * COPIES): run(i)
                e1.record(); torch.cuda.synchronize()
                print(f"M={M} {name:8s} tile={tile} split={split:2d}: {e0.elapsed_time(e1) / (10 * COPIES) * 1e3:7.1f} us (GEMM + slab reduction)", flush=True)
    sys.exit(0)
for M in (128, 256):
    tot = 0.0
    for name, N, K, sw in shapes:
        g = torch.Generator().manual_seed(1)
        a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        ws_ = [B.pack_decode_weights((torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV)) for _ in range(COPIES)]
        split = sk5(N, K)
        wsk = torch.empty(split * M * N, device=DEV) if split > 1 else None
        out = torch.empty(M, N // 2 if sw else N, dtype=torch.bfloat16, device=DEV)
        def run(i): B.gemm(a, ws_[i % COPIES], out, swiglu=sw, tile=5, split_k=split, workspace=wsk, M=M, N=N)
        for i in range(COPIES): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10 * COPIES): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (10 * COPIES) * 1e3
        tot += us
        print(f"M={M} {name:8s} N={N} K={K} split={split}: {us:7.1f} us  {N*K*2/us/1e6:5.2f} TB/s of weights")
    print(f"M={M}: {tot:.1f} us per layer = {tot/M:.3f} us per row")

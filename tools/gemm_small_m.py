"""Which tile (and split-K) serves the GEMMs of ONE utterance best?  Batch-1 prefill (M = 376) and batch-1 Whisper (M = 1500) shapes,
every legal (tile, split) pair, rotating weight copies; prints microseconds per call incl. the slab reduction, and what
icl_gemm_select_tile picks today.  usage: python tools/gemm_small_m.py [M_prefill]"""
import os, sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
lib = B.load_library()
DEV, COPIES = "cuda", 4
Mp = int(sys.argv[1]) if len(sys.argv) > 1 else 376
shapes = [("llama qkv", Mp, 12288, 4160, "bf16"), ("llama o", Mp, 4096, 4096, "res"), ("llama gate/up", Mp, 22016, 4096, "swiglu"),
          ("llama down", Mp, 4096, 11008, "res"), ("whisper qkv", 1500, 3840, 1280, "bf16"), ("whisper o", 1500, 1280, 1280, "res"),
          ("whisper fc1", 1500, 5120, 1280, "gelu"), ("whisper fc2", 1500, 1280, 5120, "res")]
for name, M, N, K, kind in shapes:
    g = torch.Generator().manual_seed(1)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    ws_ = [(torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(DEV) for _ in range(COPIES)]
    res = torch.randn(M, N, device=DEV) if kind == "res" else None
    out = torch.empty(M, N // 2 if kind == "swiglu" else N, dtype=torch.float32 if kind == "res" else torch.bfloat16, device=DEV)
    auto = lib.icl_gemm_select_tile(M, N, K, 1, 1)
    best = None
    line = []
    for tile in (1, 2, 3):
        for split in (1, 2, 3, 4, 6, 8):
            if split > 1 and (K // split < 512 or (tile == 3 and K // split < 128)):
                continue
            wsk = torch.empty(split * M * N, device=DEV) if split > 1 else None
            def run(i):
                B.gemm(a, ws_[i % COPIES], out, swiglu=kind == "swiglu", gelu=kind == "gelu", residual=res, tile=tile, split_k=split, workspace=wsk)
            try:
                for i in range(COPIES): run(i)
            except Exception as e:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(5 * COPIES): run(i)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (5 * COPIES) * 1e3
            line.append(f"t{tile}/s{split}:{us:6.1f}")
            if best is None or us < best[0]:
                best = (us, tile, split)
    print(f"{name:14s} M={M} N={N} K={K} auto=tile{auto}  best t{best[1]}/s{best[2]} {best[0]:.1f} us ({2*M*N*K/best[0]/1e6:.0f} TF/s) | " + " ".join(line), flush=True)

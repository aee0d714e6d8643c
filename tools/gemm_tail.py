"""Staircase probe for gemm256: time one shape family while M walks across a multiple of 256 workgroups (one round of the chip).
usage: python tools/gemm_tail.py  -> per shape, M-tiles vs us and TF/s.  Flat time across a round = tail quantisation."""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_speech_text_llm_amd.runtime import binding as B

lib = B.load_library()
dev = "cuda"
SHAPES = [("ll o  res->f32", 4096, 4096, 1, range(176, 209, 4)),      # 16 N-tiles: 176..208 M-tiles = 11.0 .. 13.0 rounds
          ("ll down res->f32", 4096, 11008, 1, range(176, 209, 4)),
          ("wh o  bias+res->f32", 1280, 1280, 1, range(230, 270, 5)),  # 5 N-tiles: 4.5 .. 5.3 rounds
          ("ll qkv ->bf16", 12288, 4160, 0, range(184, 198, 2))]       # 48 N-tiles
for name, N, K, res, mts in SHAPES:
    for mt in mts:
        M = mt * 256
        torch.manual_seed(0)
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.float32 if res else torch.bfloat16, device=dev)
        r = torch.randn(M, N, device=dev) if res else None
        g = B.GemmArgs()
        g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
        g.bias, g.R, g.workspace = 0, (r.data_ptr() if res else 0), 0
        g.lda, g.ldw, g.ldc, g.ldr = K, K, N, (N if res else 0)
        g.M, g.N, g.K, g.batch = M, N, K, 1
        g.epilogue = 4 if res else 0
        g.out_dtype = B.ICL_F32 if res else B.ICL_BF16
        g.res_dtype, g.split_k, g.tile = B.ICL_F32, 1, 3
        s = torch.cuda.current_stream().cuda_stream
        ts = []
        for it in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                assert lib.icl_gemm_bf16(ctypes.byref(g), s) == 0
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 3 * 1e-3)
        t = statistics.median(ts[1:])
        tiles = mt * ((N + 255) // 256)
        print(f"{name:22s} M-tiles={mt:4d} tiles={tiles:6d} rounds={tiles / 256:6.2f}  {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s  us/round={t * 1e6 / (tiles / 256):7.1f}", flush=True)
        del a, w, out, r

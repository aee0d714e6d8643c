"""A/B of libicl_hip builds on the bench's own GEMM shapes, interleaved rounds in ONE process (guide rule 24).

usage: python tools/gemm_ab.py [--rounds 5] [--iters 4] [--shapes whisper,llama,beats] name=path/to/lib.so [name=...]
Every library is loaded side by side through ctypes; per shape each round runs every library back to back, and the table
reports the median (and best) TFLOP/s per library.  With ONE library it is a plain per-shape rate table.  Outputs of the
first two libraries are also compared bit for bit (the epilogues must not change results)."""
import argparse, ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icl_speech_text_llm_amd.runtime import binding as B

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--shapes", default="whisper,llama,beats")
ap.add_argument("--scale-m", type=float, default=1.0, help="scale every M (e.g. 0.5 = micro-batch 128 shapes)")
ap.add_argument("--blaslt", action="store_true", help="also time torch.mm (hipBLASLt) on the same operands, no epilogue: a "
                "known-good reference for the main loop (guide rule 10), NOT part of the product")
ap.add_argument("libs", nargs="+")
args = ap.parse_args()

libs = []
for spec in args.libs:      # name=path[@ENV=value,...]: environment knobs a build reads once, at its first call (ICL_GEMM_GROUP_M)
    name, path = spec.split("=", 1) if "=" in spec else (os.path.basename(spec), spec)
    path, _, envs = path.partition("@")
    for kv in filter(None, envs.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.icl_gemm_bf16.restype = ctypes.c_int
    lib.icl_gemm_bf16.argtypes = [ctypes.POINTER(B.GemmArgs), ctypes.c_void_p]
    lib.icl_last_error.restype = ctypes.c_char_p
    if envs:                 # make the build read its knobs now, while they are set: one tiny call
        _a = torch.zeros(256, 128, dtype=torch.bfloat16, device="cuda"); _c = torch.empty(256, 256, dtype=torch.bfloat16, device="cuda")
        _g = B.GemmArgs(); _g.A, _g.W, _g.C = _a.data_ptr(), _a.data_ptr(), _c.data_ptr()
        _g.lda = _g.ldw = 128; _g.ldc = 256; _g.M = _g.N = 256; _g.K = 128; _g.batch = 1; _g.split_k = 1; _g.tile = 1
        _g.out_dtype, _g.res_dtype = B.ICL_BF16, B.ICL_F32
        assert lib.icl_gemm_bf16(ctypes.byref(_g), torch.cuda.current_stream().cuda_stream) == 0
        torch.cuda.synchronize()
        for kv in filter(None, envs.split(",")):
            os.environ.pop(kv.split("=")[0], None)
    libs.append((name, lib))

# (name, M, N, K, bias, gelu, residual(f32, in place), swiglu, out_f32)
SHAPES = {
    "whisper": [("wh qkv   bias->bf16", 384000, 3840, 1280, 1, 0, 0, 0, 0),
                ("wh o     bias+res->f32", 384000, 1280, 1280, 1, 0, 1, 0, 1),
                ("wh fc1   bias+gelu->bf16", 384000, 5120, 1280, 1, 1, 0, 0, 0),
                ("wh fc2   bias+res->f32", 384000, 1280, 5120, 1, 0, 1, 0, 1)],
    "llama": [("ll qkv   ->bf16 (no rope)", 48128, 12288, 4160, 0, 0, 0, 0, 0),
              ("ll o     res->f32", 48128, 4096, 4096, 0, 0, 1, 0, 1),
              ("ll gu    swiglu->bf16", 48128, 22016, 4096, 0, 0, 0, 1, 0),
              ("ll down  res->f32", 48128, 4096, 11008, 0, 0, 1, 0, 1)],
    "small": [("pf1 qkv  ->bf16", 376, 12288, 4160, 0, 0, 0, 0, 0),      # prefill of ONE 376-position prompt (no fused RoPE here)
              ("pf1 o    res->f32", 376, 4096, 4096, 0, 0, 1, 0, 1),
              ("pf1 gu   swiglu->bf16", 376, 22016, 4096, 0, 0, 0, 1, 0),
              ("pf1 down res->f32", 376, 4096, 11008, 0, 0, 1, 0, 1),
              ("wh1 qkv  bias->bf16", 1500, 3840, 1280, 1, 0, 0, 0, 0),   # one 30 s clip
              ("wh1 fc1  bias+gelu", 1500, 5120, 1280, 1, 1, 0, 0, 0),
              ("wh1 fc2  bias+res->f32", 1500, 1280, 5120, 1, 0, 1, 0, 1),
              ("be1 qkv  bias->bf16", 1496, 2304, 768, 1, 0, 0, 0, 0),
              ("be1 fc2  bias->f32", 1496, 768, 3072, 1, 0, 0, 0, 1),
              ("qf  kv   bias->bf16", 1500, 1536, 2048, 1, 0, 0, 0, 0)],
    "beats": [("be qkv   bias->bf16", 382976, 2304, 768, 1, 0, 0, 0, 0),
              ("be o     bias->f32", 382976, 768, 768, 1, 0, 0, 0, 1),
              ("be fc1   bias+gelu->bf16", 382976, 3072, 768, 1, 1, 0, 0, 0),
              ("be fc2   bias->f32", 382976, 768, 3072, 1, 0, 0, 0, 1)],
}
dev = "cuda"
stream = lambda: torch.cuda.current_stream().cuda_stream


def run(lib, g):
    rc = lib.icl_gemm_bf16(ctypes.byref(g), stream())
    if rc != 0:
        raise RuntimeError(lib.icl_last_error().decode())


for fam in args.shapes.split(","):
    for (name, M, N, K, bias, gelu, res, swiglu, of32) in SHAPES[fam]:
        M = (int(M * args.scale_m) // 256 * 256 or 256) if fam != "small" else M
        torch.manual_seed(0)
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        nout = N // 2 if swiglu else N
        outs = [torch.empty(M, nout, dtype=torch.float32 if of32 else torch.bfloat16, device=dev) for _ in libs[:2]]
        outs += [outs[-1]] * max(0, len(libs) - 2)
        b = torch.randn(N, device=dev) if bias else None
        r = torch.randn(M, nout, device=dev) if res else None
        gs = []
        for li in range(len(libs)):
            g = B.GemmArgs()
            g.A, g.W, g.C = a.data_ptr(), w.data_ptr(), outs[li].data_ptr()
            g.bias, g.R, g.workspace = (b.data_ptr() if bias else 0), (r.data_ptr() if res else 0), 0
            g.lda, g.ldw, g.ldc, g.ldr = K, K, nout, (nout if res else 0)
            g.strideA = g.strideC = g.strideR = 0
            g.M, g.N, g.K, g.batch = M, N, K, 1
            g.epilogue = (1 if bias else 0) | (2 if gelu else 0) | (4 if res else 0) | (8 if swiglu else 0)
            g.out_dtype = B.ICL_F32 if of32 else B.ICL_BF16
            g.res_dtype, g.split_k, g.tile = B.ICL_F32, 1, (3 if fam != "small" else 0)
            gs.append(g)
        for (nm, lib), g in zip(libs, gs):          # warm-up + result check
            run(lib, g)
        torch.cuda.synchronize()
        same = torch.equal(outs[0], outs[1]) if len(libs) > 1 else None
        times = {nm: [] for nm, _ in libs}
        for _ in range(args.rounds):
            for (nm, lib), g in zip(libs, gs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    run(lib, g)
                e1.record()
                torch.cuda.synchronize()
                times[nm].append(e0.elapsed_time(e1) / args.iters * 1e-3)
        if args.blaslt:
            wt = w.t()
            torch.mm(a, wt)
            torch.cuda.synchronize()
            times["torch.mm"] = []
            for _ in range(args.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    torch.mm(a, wt)
                e1.record()
                torch.cuda.synchronize()
                times["torch.mm"].append(e0.elapsed_time(e1) / args.iters * 1e-3)
        fl = 2.0 * M * N * K
        cells = "  ".join(f"{nm}: {fl / statistics.median(t) / 1e12:7.1f} (best {fl / min(t) / 1e12:7.1f}) TF/s {statistics.median(t) * 1e6:8.1f} us"
                          for nm, t in times.items())
        print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d}  {cells}" + (f"  bit-equal={same}" if same is not None else ""), flush=True)
        del a, w, outs, b, r
        torch.cuda.empty_cache()

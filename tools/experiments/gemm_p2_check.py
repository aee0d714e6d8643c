"""Persistent 256x256 GEMM (tile 7) against the one-tile-per-workgroup kernel (tile 3): bit identity over epilogue forms and shapes,
repeated launches (race screen), and timing at the bench's shapes."""
import sys, os, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
B.load_library()
DEV = "cuda"
torch.manual_seed(0)

def run(tile, a, w, out, **kw):
    B.gemm(a, w, out, tile=tile, **kw)
    return out

def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def case(name, M, N, K, form, reps=6, timing=False):
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    kw, odt, ncol = {}, torch.bfloat16, N
    if form == "bf16": pass
    elif form == "bias": kw = dict(bias=bias)
    elif form == "gelu": kw = dict(bias=bias, gelu=True)
    elif form == "swiglu": kw, ncol = dict(swiglu=True), N // 2
    elif form == "f32": odt = torch.float32; kw = dict(bias=bias)
    elif form == "res": odt = torch.float32
    ok = True
    def outputs(tile):
        if form == "res":      # in place on the f32 residual stream, as the layers do
            h = r0.clone()
            run(tile, a, w, h, bias=bias, residual=h)
            return h
        o = torch.full((M, ncol), float("nan"), dtype=odt, device=DEV)
        run(tile, a, w, o, **kw)
        return o
    r0 = torch.randn(M, N, device=DEV) if form == "res" else None
    ref = outputs(3)
    for i in range(reps):
        got = outputs(7)
        if not torch.equal(got, ref):
            bad = (got != ref) | (got.isnan() != ref.isnan())
            rows = bad.any(1).nonzero().flatten()
            print(f"  MISMATCH {name} rep {i}: {int(bad.sum())} elements, first rows {rows[:6].tolist()}")
            ok = False
            break
    msg = f"{name:34s} M={M:6d} N={N:5d} K={K:5d} {form:7s}: {'bit-identical x%d' % reps if ok else 'FAILED'}"
    if timing and ok:
        if form == "res":
            h = r0.clone()
            t3 = timeit(lambda: run(3, a, w, h, bias=bias, residual=h))
            t7 = timeit(lambda: run(7, a, w, h, bias=bias, residual=h))
        else:
            o = torch.empty(M, ncol, dtype=odt, device=DEV)
            t3 = timeit(lambda: run(3, a, w, o, **kw))
            t7 = timeit(lambda: run(7, a, w, o, **kw))
        fl = 2.0 * M * N * K
        msg += f"   tile3 {t3*1e6:8.1f} us {fl/t3/1e12:7.1f} TF/s | tile7 {t7*1e6:8.1f} us {fl/t7/1e12:7.1f} TF/s  ({(t3/t7-1)*100:+.1f} %)"
    print(msg, flush=True)
    return ok

allok = True
# small / odd tile counts first (fewer tiles than CUs, exactly CUs, CUs + 1, several rounds, K at the lower bound)
for (M, N, K) in [(256, 256, 256), (512, 768, 320), (2048, 2048, 256), (4096, 4096, 512), (4352, 4096, 384), (8192, 5120, 1280), (6144, 2816, 768)]:
    for form in ("bf16", "bias", "gelu", "swiglu", "f32", "res"):
        allok &= case("small", M, N, K, form, reps=4)
if len(sys.argv) > 1 and sys.argv[1] == "bench":
    for (name, M, N, K, form) in [("whisper qkv", 192000, 3840, 1280, "bias"), ("whisper fc1", 192000, 5120, 1280, "gelu"),
                                   ("whisper fc2", 192000, 1280, 5120, "res"), ("whisper o", 192000, 1280, 1280, "res"),
                                   ("llama gate/up", 48128, 22016, 4096, "swiglu"), ("llama down", 48128, 4096, 11008, "res"),
                                   ("llama o", 48128, 4096, 4096, "res"), ("beats fc1", 191488, 3072, 768, "gelu")]:
        allok &= case(name, M, N, K, form, reps=3, timing=True)
print("ALL OK" if allok else "FAILURES")

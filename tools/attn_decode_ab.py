"""Time icl_attn_decode_bf16 at small and large batches for one or more builds (name=path ...), interleaved, and compare outputs.
usage: python tools/attn_decode_ab.py base=lib/libicl_hip_base.so new=lib/libicl_hip.so"""
import ctypes, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = []
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.icl_attn_decode_bf16.restype = ctypes.c_int
    lib.icl_attn_decode_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                         ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                         ctypes.c_void_p]
    libs.append((name, lib))
H, D, L = 32, 128, 448
dev = "cuda"
for Bn in (1, 4, 16, 64, 256):
    torch.manual_seed(Bn)
    q = torch.randn(Bn, H * D, device=dev).to(torch.bfloat16)
    kc = torch.randn(Bn, H, L, D, device=dev).to(torch.bfloat16)
    vc = torch.randn(Bn, H, L, D, device=dev).to(torch.bfloat16)
    lens = torch.full((Bn,), 386, dtype=torch.int32, device=dev)
    outs = [torch.empty(Bn, H * D, dtype=torch.bfloat16, device=dev) for _ in libs]
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for (name, lib), o in zip(libs, outs):
        call = lambda: lib.icl_attn_decode_bf16(q.data_ptr(), H * D, kc.data_ptr(), vc.data_ptr(), o.data_ptr(), H * D, lens.data_ptr(), Bn, H, D, L,
                                                D ** -0.5, s)
        for _ in range(5):
            assert call() == 0
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                call()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        res[name] = statistics.median(ts)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    print(f"B={Bn:4d}  " + "  ".join(f"{n}: {t:7.1f} us" for n, t in res.items()) + f"  bit-equal={same}", flush=True)

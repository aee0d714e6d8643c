"""Phase stamps (s_memtime) of the 256x256 GEMM from the diagnostic build lib/libicl_hip_stamp.so (tools only; never shipped)."""
import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
B.LIB_PATH = "icl-speech-text-llm_amd/lib/libicl_hip_stamp.so"
B.load_library()
DEV = "cuda"
for (name, M, N, K, kw) in [("whisper qkv (bias)", 192000, 3840, 1280, "bias"), ("llama qkv-like bf16", 48128, 12288, 4096, "none"),
                            ("whisper fc1 (gelu)", 192000, 5120, 1280, "gelu")]:
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    nblk = (M // 256) * (N // 256)
    ws = torch.zeros(nblk * 8 * 8 * 2, dtype=torch.float32, device=DEV)     # 8 waves x 8 u64 per block
    args = dict(bias=bias) if kw != "none" else {}
    if kw == "gelu": args["gelu"] = True
    for _ in range(3):
        B.gemm(a, w, out, tile=3, workspace=ws, **args)
    torch.cuda.synchronize()
    st = ws.cpu().numpy().view(np.uint64).reshape(nblk, 8, 8).astype(np.float64)
    d = st[:, :, 1:6] - st[:, :, 0:5]                      # phase durations in shader cycles, per block and wave
    tot = st[:, :, 5] - st[:, :, 0]
    clk = (st[:, :, 5] - st[:, :, 0]) / np.maximum((st[:, :, 7] - st[:, :, 6]), 1) * 100.0     # MHz: memtime / memrealtime(100 MHz)
    names = ["entry->prologue issued", "prologue issued->first data + barrier", "main loop", "C -> LDS (+GELU) + barriers", "row reads + stores issued"]
    print(f"{name}: {nblk} blocks, K-tiles {K//64}; median cycles per phase (wave 0), clock ~{np.median(clk):.0f} MHz")
    for i, n in enumerate(names):
        print(f"   {n:42s} {np.median(d[:, 0, i]):9.0f}  ({100*np.median(d[:, 0, i])/np.median(tot[:, 0]):4.1f} %)   p90 {np.percentile(d[:, 0, i], 90):9.0f}")
    print(f"   {'total per tile':42s} {np.median(tot[:, 0]):9.0f}  = {np.median(tot[:,0])/np.median(clk)*1e0:6.2f} us at that clock; main loop per K-tile {np.median(d[:,0,2])/(K//64):6.0f} cycles")
    # launch-level: first entry to last exit (realtime, 100 MHz)
    span = (st[:, :, 7].max() - st[:, :, 6].min()) / 100.0
    print(f"   launch span {span:8.1f} us; sum of tile times per CU slot ~ {np.sum(tot[:,0])/256/np.median(clk):8.1f} us")

"""Which kernel does hipBLASLt (torch.mm) pick on the bench's GEMM shapes?  Run under `rocprofv3 --kernel-trace`: the Tensile
kernel name spells its macro tile, MFMA shape, wave layout and staging options; the trace adds VGPRs / LDS / workgroup size.
A reference for the main loop only (guide rule 10) — nothing of the product calls a BLAS library."""
import torch
for (M, N, K) in ((48128, 4096, 11008), (48128, 22016, 4096), (384000, 3840, 1280), (382976, 3072, 768)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    for _ in range(3):
        torch.mm(a, w.t())
    torch.cuda.synchronize()
    del a, w

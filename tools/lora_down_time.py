"""Times icl_lora_down_bf16 at the decode shape (128 rows, K = 4096, r = 16, rotating A matrices); prints a checksum."""
import hashlib, os, sys
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
B.load_library()
g = torch.Generator().manual_seed(3)
M, K0, r, L = 128, 4096, 16, 32
x = torch.randn(M, K0 + 64, generator=g).to(torch.bfloat16).cuda()
As = [torch.randn(r, K0, generator=g).to(torch.bfloat16).cuda() for _ in range(L)]
for a in As:
    B.lora_down(x, K0, a, r, 2.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    for a in As:
        B.lora_down(x, K0, a, r, 2.0)
e1.record()
torch.cuda.synchronize()
h = hashlib.sha256(x.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
print(f"lora_down: {e0.elapsed_time(e1) / (20 * L) * 1e3:.2f} us  sha {h}")

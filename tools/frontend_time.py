"""Times the two audio front-ends (K1 Whisper log-mel, K4 Kaldi fbank) alone at the bench's 128 clips x 30 s, and prints a
checksum of their outputs (ICL_LIB=<other build> for a same-box A/B; the checksums must agree bit for bit)."""
import hashlib
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
from icl_speech_text_llm_amd.runtime import audio_tables as af

if os.environ.get("ICL_LIB"):
    B.LIB_PATH = os.environ["ICL_LIB"]
B.load_library()
DEV = "cuda"
n = 128
g = torch.Generator().manual_seed(5)
wav = (torch.randn(n, 480000, generator=g) * 0.1).clamp(-1, 1).to(DEV)
wl = torch.full((n,), 480000, dtype=torch.int32, device=DEV)
wl[3] = 16000 * 7 + 123
mel = torch.from_numpy(af.slaney_mel_filters(80)).to(DEV)
banks = torch.from_numpy(af.kaldi_mel_banks()).to(DEV)
spec = torch.empty(n, 80, 3000, device=DEV)
xt = torch.empty(n, 3002, 80, dtype=torch.bfloat16, device=DEV)
ws = torch.empty(n * 80 * 3000 + n, dtype=torch.float32, device=DEV)
max_frames = 2998
fb = torch.zeros(n, max_frames, 128, device=DEV)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t1 = timed(lambda: B.logmel_whisper(wav, wl, mel, 80, spec, xt, ws))
t2 = timed(lambda: B.fbank_kaldi(wav, wl, banks, max_frames, 15.41663, 6.55582, fb))
h = hashlib.sha256(spec.cpu().numpy().tobytes() + xt.view(torch.int16).cpu().numpy().tobytes() + fb.cpu().numpy().tobytes()).hexdigest()[:16]
print(f"frontends: logmel {t1:.3f} ms  fbank {t2:.3f} ms  sha {h}")

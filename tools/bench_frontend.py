"""K1 log-mel and K4 Kaldi fbank at the bench shape (128 clips of 30 s)."""
import sys, os, torch
sys.path.insert(0, ".")
from icl_speech_text_llm_amd.runtime import binding as B
from icl_speech_text_llm_amd.runtime.engines import LogMel, Workspace
from icl_speech_text_llm_amd.runtime.audio_tables import kaldi_mel_banks
DEV = "cuda"
B.load_library()
n = 128
wav = (torch.randn(n, 480000, device=DEV) * 0.1).clamp(-1, 1)
lens = torch.full((n,), 480000, dtype=torch.int32, device=DEV)
ws, lm = Workspace(torch.device(DEV)), LogMel(80, DEV)
banks = torch.from_numpy(kaldi_mel_banks()).to(DEV)
fb = torch.empty(n, 2998, 128, device=DEV)
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
print(f"log-mel {t(lambda: lm(ws, wav, lens)):.2f} ms   fbank {t(lambda: B.fbank_kaldi(wav, lens, banks, 2998, 15.41663, 6.55582, fb)):.2f} ms")

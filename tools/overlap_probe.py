"""Probe: does the decode phase (HBM-bound, a captured graph) hide behind the speech encoders (MFMA-bound) when the two run on
separate streams?  Sequential vs concurrent wall time of [encode_speech(256 clips)] and [decode graph replay of 256 sequences].
Diagnostic only — the runtime runs them back to back."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from icl_speech_text_llm_amd.runtime import synth
from icl_speech_text_llm_amd.runtime.config import SalmonnCfg
from icl_speech_text_llm_amd.runtime.salmonn import SalmonnRuntime

dev = torch.device("cuda:0")
cfg = SalmonnCfg.llama2_7b()
sd = synth.salmonn_state(cfg, seed=0, device=dev, dtype=torch.bfloat16)
rt = SalmonnRuntime(cfg, dict(sd), device=dev)
del sd
Bm = 256
w, ids = bench.synth_utterances(0, Bm, cfg.llama.vocab)
wav = torch.from_numpy(w).to(dev)
prompts = bench.build_prompts(ids)
lens = [480000] * Bm
for _ in range(3):                                   # eager, capture, replay
    speech = rt.encode_speech(wav, lens)
    rt.generate(prompts, speech, max_new_tokens=10, suppress_eos=True)
torch.cuda.synchronize()
graph = next(iter(rt._graphs.values()))
side = torch.cuda.Stream(device=dev)

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

t_enc = timed(lambda: rt.encode_speech(wav, lens))
t_dec = timed(lambda: graph.replay())
def both():
    with torch.cuda.stream(side):
        graph.replay()
    rt.encode_speech(wav, lens)
t_both = timed(both)
print(f"overlap probe: encoders {t_enc:.1f} ms, decode graph {t_dec:.1f} ms, sum {t_enc + t_dec:.1f} ms, concurrent {t_both:.1f} ms")

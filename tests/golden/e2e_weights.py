"""Weights of the end-to-end goldens, rebuilt from a seed by the generator (tests/golden/make_golden.py) and by the GPU test
alike: a miniature SALMONN (this build's `tiny` arch with BEATs, no LoRA) whose matrices are bf16-representable, with a
Llama whose layer projections are small next to its embeddings / LM head — the bf16 rounding inside the layers then moves
the logits by far less than the typical top-1 / top-2 gap, so that a seed with decisive greedy margins exists."""
import torch


def tiny_salmonn_weights(seed: int):
    from icl_speech_text_llm_amd.runtime import synth
    from icl_speech_text_llm_amd.runtime.config import SalmonnCfg
    cfg = SalmonnCfg.tiny(use_beats=True, lora=False)
    sd = synth.salmonn_state(cfg, seed=seed, jitter=True)
    g = torch.Generator().manual_seed(10_000 + seed)
    out = {}
    for k, v in sd.items():
        if k.startswith("llama_model."):
            if v.dim() > 1:
                v = torch.randn(v.shape, generator=g) * (0.03 if "_proj" in k else 0.3)
            else:
                v = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        if v.is_floating_point() and v.dim() > 1:
            v = v.to(torch.bfloat16).float()
        out[k] = v
    return cfg, out

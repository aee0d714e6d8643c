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


def tiny_qwen_hf_model(seed: int):
    """HF Qwen2AudioForConditionalGeneration at this build's `QwenAudioCfg.tiny()` dims with seeded, bf16-representable weights
    (decoder layer projections small, as above); generation config: greedy, eos 2, pad 299 — what a checkpoint folder would say."""
    from transformers import Qwen2AudioConfig, Qwen2AudioEncoderConfig, Qwen2AudioForConditionalGeneration, Qwen2Config
    from icl_speech_text_llm_amd.runtime.config import QwenAudioCfg
    from icl_speech_text_llm_amd.runtime.synth import whisper_sinusoids
    c = QwenAudioCfg.tiny(lora=False)
    acfg = Qwen2AudioEncoderConfig(num_mel_bins=c.audio.n_mels, d_model=c.audio.d_model, encoder_layers=c.audio.n_layers,
                                   encoder_attention_heads=c.audio.n_heads, encoder_ffn_dim=c.audio.ffn, max_source_positions=1500)
    tcfg = Qwen2Config(vocab_size=c.llm.vocab, hidden_size=c.llm.hidden, intermediate_size=c.llm.ffn, num_hidden_layers=c.llm.n_layers,
                       num_attention_heads=c.llm.n_heads, num_key_value_heads=c.llm.n_heads, rms_norm_eps=c.llm.rms_eps,
                       rope_theta=c.llm.rope_theta, max_position_embeddings=c.llm.max_pos, tie_word_embeddings=False)
    m = Qwen2AudioForConditionalGeneration(Qwen2AudioConfig(audio_config=acfg, text_config=tcfg, audio_token_index=c.audio_token_id)).eval()
    g = torch.Generator().manual_seed(20_000 + seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "embed_positions" in n:
                p.copy_(whisper_sinusoids(1500, c.audio.d_model))
                continue
            if p.dim() > 1:
                lm = "language_model" in n or n.startswith("lm_head")
                std = (0.03 if "_proj" in n else 0.3) if lm else 0.08
                v = (torch.randn(p.shape, generator=g) * std).to(torch.bfloat16).float()
            else:
                v = (1.0 if ("norm" in n and n.endswith("weight")) else 0.0) + 0.1 * torch.randn(p.shape, generator=g)
            p.copy_(v)
    m.generation_config.do_sample = False
    m.generation_config.eos_token_id, m.generation_config.pad_token_id = c.llm.eos_id, c.llm.pad_id
    return c, m


def qwen_batch(cfg, seed: int = 0):
    """One prompt with two audios (closed-form features, 3000 and 1234 valid mel frames) and a 4-token completion."""
    c, t = torch.arange(128.0)[:, None], torch.arange(3000.0)[None, :]
    feats = torch.stack([0.5 * torch.sin(0.01 * (c + 1.0) * t + c), 0.4 * torch.cos(0.013 * (c + 2.0) * t)])
    mel_lens = [3000, 1234]
    fmask = torch.zeros(2, 3000, dtype=torch.long)
    for i, L in enumerate(mel_lens):
        fmask[i, :L] = 1
    feats = feats * fmask[:, None, :]
    outl = [((L - 1) // 2 + 1 - 2) // 2 + 1 for L in mel_lens]
    g = torch.Generator().manual_seed(30_000 + seed)
    txt = lambda k: torch.randint(3, cfg.audio_token_id - 8, (k,), generator=g).tolist()      # noqa: E731
    prompt = txt(10) + [cfg.audio_token_id] * outl[0] + txt(5) + [cfg.audio_token_id] * outl[1] + txt(6)
    full = prompt + txt(4)
    ids = torch.tensor([full])
    return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "input_features": feats, "feature_attention_mask": fmask,
            "prompt_length": torch.tensor([len(prompt)])}, len(prompt)

"""Seeded synthetic checkpoints under the reference stack's ``state_dict`` key names.

There is no network here, so neither ``salmonn_v1.pth`` nor the HF checkpoints the reference loads
(models/custom_salmon.py:30-32,47) exist; SURVEY.md §8(d) freezes the substitute: weights
``N(0, 0.02^2)``, norm gains 1, LoRA ``B = 0.01*N(0,1)``, one ``torch.Generator`` seed.  The same dict
feeds the HIP path (through runtime/packing.py) and the CPU oracle, so parity is tested on identical
weights.  Key names are the ones a real SALMONN checkpoint uses (``speech_encoder.*``, ``beats.*``,
``speech_Qformer.bert.*``, ``speech_query_tokens``, ``speech_llama_proj.*``, ``ln_speech``/``ln_audio``,
``llama_model.*``) with the peft wrapping already normalised (see runtime/packing.normalize_keys).
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .config import BeatsCfg, LlamaCfg, QFormerCfg, SalmonnCfg, WhisperCfg

SD = Dict[str, torch.Tensor]


class _Gen:
    def __init__(self, seed: int, device, dtype, jitter: bool):
        self.device = torch.device(device)
        self.g = torch.Generator(device=self.device).manual_seed(seed)
        self.dtype = dtype
        self.jitter = jitter

    def normal(self, *shape, std=0.02):
        # generated in f32 chunks to bound peak memory, stored in the checkpoint dtype
        out = torch.empty(*shape, dtype=self.dtype, device=self.device)
        flat = out.view(-1)
        step = 1 << 26
        for i in range(0, flat.numel(), step):
            n = min(step, flat.numel() - i)
            flat[i:i + n] = (torch.randn(n, generator=self.g, device=self.device) * std).to(self.dtype)
        return out

    def gain(self, n):
        base = torch.ones(n, dtype=torch.float32, device=self.device)
        if self.jitter:
            base = base + 0.1 * torch.randn(n, generator=self.g, device=self.device)
        return base

    def bias(self, n):
        if self.jitter:
            return 0.02 * torch.randn(n, generator=self.g, device=self.device)
        return torch.zeros(n, dtype=torch.float32, device=self.device)


def whisper_sinusoids(length: int, channels: int) -> torch.Tensor:
    inc = math.log(10000.0) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length)[:, None] * inv[None, :]
    return torch.cat([t.sin(), t.cos()], dim=1)


def whisper_state(cfg: WhisperCfg, g: _Gen, prefix: str = "speech_encoder.") -> SD:
    d, sd = cfg.d_model, {}
    sd[prefix + "conv1.weight"] = g.normal(d, cfg.n_mels, 3)
    sd[prefix + "conv1.bias"] = g.bias(d)
    sd[prefix + "conv2.weight"] = g.normal(d, d, 3)
    sd[prefix + "conv2.bias"] = g.bias(d)
    sd[prefix + "embed_positions.weight"] = (whisper_sinusoids(cfg.n_ctx, d) * (0.1 if g.jitter else 1.0)).to(g.device)
    for i in range(cfg.n_layers):
        lp = f"{prefix}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[lp + f"self_attn.{n}.weight"] = g.normal(d, d)
            if n != "k_proj":
                sd[lp + f"self_attn.{n}.bias"] = g.bias(d)
        sd[lp + "self_attn_layer_norm.weight"], sd[lp + "self_attn_layer_norm.bias"] = g.gain(d), g.bias(d)
        sd[lp + "fc1.weight"], sd[lp + "fc1.bias"] = g.normal(cfg.ffn, d), g.bias(cfg.ffn)
        sd[lp + "fc2.weight"], sd[lp + "fc2.bias"] = g.normal(d, cfg.ffn), g.bias(d)
        sd[lp + "final_layer_norm.weight"], sd[lp + "final_layer_norm.bias"] = g.gain(d), g.bias(d)
    sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"] = g.gain(d), g.bias(d)
    return sd


def beats_state(cfg: BeatsCfg, g: _Gen, prefix: str = "beats.") -> SD:
    d, sd = cfg.d_model, {}
    sd[prefix + "patch_embedding.weight"] = g.normal(cfg.embed, 1, 16, 16)
    sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"] = g.gain(cfg.embed), g.bias(cfg.embed)
    sd[prefix + "post_extract_proj.weight"], sd[prefix + "post_extract_proj.bias"] = g.normal(d, cfg.embed), g.bias(d)
    cpg = d // cfg.conv_groups
    v = g.normal(d, cpg, cfg.conv_pos)
    sd[prefix + "encoder.pos_conv.0.weight_v"] = v
    sd[prefix + "encoder.pos_conv.0.weight_g"] = v.float().norm(dim=(0, 1), keepdim=True) * (1.0 + (0.1 if g.jitter else 0.0))
    sd[prefix + "encoder.pos_conv.0.bias"] = g.bias(d)
    sd[prefix + "encoder.layer_norm.weight"], sd[prefix + "encoder.layer_norm.bias"] = g.gain(d), g.bias(d)
    sd[prefix + "encoder.layers.0.self_attn.relative_attention_bias.weight"] = g.normal(cfg.num_buckets, cfg.n_heads, std=0.2).float()
    for i in range(cfg.n_layers):
        lp = f"{prefix}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[lp + f"self_attn.{n}.weight"], sd[lp + f"self_attn.{n}.bias"] = g.normal(d, d), g.bias(d)
        sd[lp + "self_attn.grep_linear.weight"] = g.normal(8, d // cfg.n_heads, std=0.2).float()
        sd[lp + "self_attn.grep_linear.bias"] = g.bias(8)
        sd[lp + "self_attn.grep_a"] = torch.ones(1, cfg.n_heads, 1, 1, device=g.device) * (1.2 if g.jitter else 1.0)
        sd[lp + "self_attn_layer_norm.weight"], sd[lp + "self_attn_layer_norm.bias"] = g.gain(d), g.bias(d)
        sd[lp + "fc1.weight"], sd[lp + "fc1.bias"] = g.normal(cfg.ffn, d), g.bias(cfg.ffn)
        sd[lp + "fc2.weight"], sd[lp + "fc2.bias"] = g.normal(d, cfg.ffn), g.bias(d)
        sd[lp + "final_layer_norm.weight"], sd[lp + "final_layer_norm.bias"] = g.gain(d), g.bias(d)
    return sd


def qformer_state(cfg: QFormerCfg, llm_hidden: int, g: _Gen, whisper_d: int, beats_d: int) -> SD:
    h, sd = cfg.hidden, {}
    p = "speech_Qformer.bert."
    sd["ln_speech.weight"], sd["ln_speech.bias"] = g.gain(whisper_d), g.bias(whisper_d)
    if beats_d:
        sd["ln_audio.weight"], sd["ln_audio.bias"] = g.gain(beats_d), g.bias(beats_d)
    sd["speech_query_tokens"] = g.normal(1, cfg.n_query, h).float()
    sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"] = g.gain(h), g.bias(h)
    for i in range(cfg.n_layers):
        lp = f"{p}encoder.layer.{i}."
        for blk, kdim in (("attention", h), ("crossattention", cfg.enc_width)):
            sd[lp + f"{blk}.self.query.weight"], sd[lp + f"{blk}.self.query.bias"] = g.normal(h, h), g.bias(h)
            sd[lp + f"{blk}.self.key.weight"], sd[lp + f"{blk}.self.key.bias"] = g.normal(h, kdim), g.bias(h)
            sd[lp + f"{blk}.self.value.weight"], sd[lp + f"{blk}.self.value.bias"] = g.normal(h, kdim), g.bias(h)
            sd[lp + f"{blk}.output.dense.weight"], sd[lp + f"{blk}.output.dense.bias"] = g.normal(h, h), g.bias(h)
            sd[lp + f"{blk}.output.LayerNorm.weight"], sd[lp + f"{blk}.output.LayerNorm.bias"] = g.gain(h), g.bias(h)
        sd[lp + "intermediate_query.dense.weight"], sd[lp + "intermediate_query.dense.bias"] = g.normal(cfg.ffn, h), g.bias(cfg.ffn)
        sd[lp + "output_query.dense.weight"], sd[lp + "output_query.dense.bias"] = g.normal(h, cfg.ffn), g.bias(h)
        sd[lp + "output_query.LayerNorm.weight"], sd[lp + "output_query.LayerNorm.bias"] = g.gain(h), g.bias(h)
    sd["speech_llama_proj.weight"], sd["speech_llama_proj.bias"] = g.normal(llm_hidden, h), g.bias(llm_hidden)
    return sd


def llama_state(cfg: LlamaCfg, g: _Gen, prefix: str = "llama_model.", margin: bool = False) -> SD:
    """``margin=True``: a decoder whose greedy decisions are DECISIVE, for token-exactness checks at full size.  Under the
    frozen N(0, 0.02^2) weights every layer adds ~2 rms of pseudo-random signal to a 0.02-rms embedding, so the final logits
    are near-degenerate Gaussians whose top-1 margin is a lottery (0.004 on the bench's first utterance) — no statement about
    arg-max exactness is possible there.  Here the embedding has unit rms, the two residual writers (o_proj, down_proj) are
    scaled by 0.05 so the residual stream stays aligned with the last token's embedding (cos ~0.9 after 32 layers), and
    ``lm_head[perm[t]] = 0.01 * embed[t]`` for a seeded permutation: the next token is perm[last token] with a margin of tens of
    logits while every kernel of the path (all GEMM tiles, RoPE, cache append, attention, norms, arg-max) still runs at size."""
    h, sd = cfg.hidden, {}
    p = prefix + "model."
    sd[p + "embed_tokens.weight"] = g.normal(cfg.vocab, h, std=1.0 if margin else 0.02)
    res_std = 0.001 if margin else 0.02
    for i in range(cfg.n_layers):
        lp = f"{p}layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[lp + f"self_attn.{n}.weight"] = g.normal(h, h, std=res_std if n == "o_proj" else 0.02)
        if cfg.qkv_bias:
            for n in ("q_proj", "k_proj", "v_proj"):
                sd[lp + f"self_attn.{n}.bias"] = g.bias(h) if g.jitter else 0.02 * torch.randn(h, generator=g.g, device=g.device)
        if cfg.lora_rank:
            for n in cfg.lora_targets:
                sd[lp + f"self_attn.{n}.lora_A.weight"] = g.normal(cfg.lora_rank, h)
                sd[lp + f"self_attn.{n}.lora_B.weight"] = g.normal(h, cfg.lora_rank, std=0.01)
        sd[lp + "mlp.gate_proj.weight"] = g.normal(cfg.ffn, h)
        sd[lp + "mlp.up_proj.weight"] = g.normal(cfg.ffn, h)
        sd[lp + "mlp.down_proj.weight"] = g.normal(h, cfg.ffn, std=res_std)
        sd[lp + "input_layernorm.weight"] = g.gain(h)
        sd[lp + "post_attention_layernorm.weight"] = g.gain(h)
    sd[p + "norm.weight"] = g.gain(h)
    if margin:
        perm = torch.randperm(cfg.vocab, generator=g.g, device=g.device)
        head = torch.empty_like(sd[p + "embed_tokens.weight"])
        head[perm] = (sd[p + "embed_tokens.weight"].float() * 0.01).to(head.dtype)
        sd[prefix + "lm_head.weight"] = head
    else:
        sd[prefix + "lm_head.weight"] = g.normal(cfg.vocab, h)
    return sd


def margin_successor(sd: SD, token: int, prefix: str = "llama_model.") -> int:
    """The token a ``margin=True`` decoder is built to emit after ``token``: arg-max_j lm_head[j] . embed[token]
    (= perm[token], since lm_head[perm[t]] = 0.01 * embed[t] and distinct rows are near-orthogonal)."""
    e = sd[prefix + "model.embed_tokens.weight"][token].float()
    return int((sd[prefix + "lm_head.weight"].float() @ e).argmax())


def salmonn_state(cfg: SalmonnCfg, seed: int = 0, device="cpu", dtype=torch.float32, jitter: bool = False,
                  parts=("whisper", "beats", "qformer", "llama"), margin: bool = False) -> SD:
    """Full synthetic SALMONN checkpoint.  ``dtype`` applies to matrices; gains/biases stay f32."""
    g = _Gen(seed, device, dtype, jitter)
    sd: SD = {}
    if "whisper" in parts:
        sd.update(whisper_state(cfg.whisper, g))
    if "beats" in parts and cfg.beats is not None:
        sd.update(beats_state(cfg.beats, g))
    if "qformer" in parts:
        sd.update(qformer_state(cfg.qformer, cfg.llama.hidden, g, cfg.whisper.d_model,
                                cfg.beats.d_model if cfg.beats is not None else 0))
    if "llama" in parts:
        sd.update(llama_state(cfg.llama, g, margin=margin))
    return sd


def qwen_audio_state(cfg, seed: int = 0, device="cpu", dtype=torch.float32, jitter: bool = False, margin: bool = False) -> SD:
    """Synthetic Qwen2-Audio checkpoint under HF Qwen2AudioForConditionalGeneration names: ``audio_tower.*``,
    ``multi_modal_projector.linear.*``, ``language_model.model.*`` / ``language_model.lm_head.weight``."""
    g = _Gen(seed, device, dtype, jitter)
    sd: SD = {}
    sd.update(whisper_state(cfg.audio, g, prefix="audio_tower."))
    sd["multi_modal_projector.linear.weight"] = g.normal(cfg.llm.hidden, cfg.audio.d_model)
    sd["multi_modal_projector.linear.bias"] = g.bias(cfg.llm.hidden)
    sd.update(llama_state(cfg.llm, g, prefix="language_model.", margin=margin))
    return sd

"""Checkpoint -> HBM layout of the HIP path.

Takes ``state_dict`` tensors under the reference stack's key names (what
``inference/inference.py:157-177`` feeds to ``load_state_dict(strict=False)``) and lays them out the
way the kernels want them resident in HBM:

* every GEMM weight as bf16 ``[N, K]`` (nn.Linear layout, K contiguous, 16-byte aligned rows);
* q/k/v fused into one ``[3h, K]`` weight (one GEMM, one pass over the activations);
* Llama gate/up fused and row-interleaved in blocks of 16 for the in-register SwiGLU epilogue;
* un-merged LoRA kept exact as a K-augmentation: ``W_aug = [W | B_q 0 / 0 0 / 0 B_v | 0]`` against
  ``x_aug = [x | s*A_q x | s*A_v x | 0]`` (K grows by 64) — same arithmetic as peft's
  ``W x + (alpha/r) B A x`` with no extra GEMM launch and no rounding of the merged weight;
* Conv1d/Conv2d weights re-ordered to the (tap-major, channel-minor) K order of the affine
  row views the GEMM reads (Whisper conv stem, BEATs patch embedding and grouped positional conv);
* biases, norm gains and the BEATs relative-position table in f32.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from .config import BeatsCfg, LlamaCfg, QFormerCfg, WhisperCfg

SD = Dict[str, torch.Tensor]
LORA_PAD = 64  # K-augmentation columns (multiple of the GEMM's BK)


def normalize_keys(sd: SD) -> SD:
    """Map peft / DDP / HF wrapper spellings onto the canonical names used here.

    ``…base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight`` (form evidenced at
    models/mlp_salmonn__.py:1065) -> ``…model.layers.0.self_attn.q_proj.lora_A.weight``; also strips
    ``module.`` (DDP), ``salmonn.`` (CustomSALMONN attribute) and peft>=0.6 ``.base_layer``."""
    out = {}
    for k, v in sd.items():
        k2 = k
        for junk in ("module.", "salmonn."):
            if k2.startswith(junk):
                k2 = k2[len(junk):]
        k2 = k2.replace("llama_model.base_model.model.", "llama_model.")
        k2 = k2.replace(".base_layer.", ".").replace(".lora_A.default.", ".lora_A.").replace(".lora_B.default.", ".lora_B.")
        out[k2] = v
    return out


def _bf(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.bfloat16).contiguous()


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _take(sd: SD, key: str, consume: bool) -> torch.Tensor:
    if key not in sd:
        raise KeyError(f"checkpoint is missing '{key}'")
    return sd.pop(key) if consume else sd[key]


# ------------------------------------------------------------------------------------------------
@dataclass
class EncLayer:            # pre-LN transformer layer (Whisper) / post-LN (BEATs): same tensors
    ln1_g: torch.Tensor
    ln1_b: torch.Tensor
    wqkv: torch.Tensor
    bqkv: torch.Tensor
    wo: torch.Tensor
    bo: torch.Tensor
    ln2_g: torch.Tensor
    ln2_b: torch.Tensor
    w1: torch.Tensor
    b1: torch.Tensor
    w2: torch.Tensor
    b2: torch.Tensor
    extra: dict = field(default_factory=dict)


@dataclass
class PackedWhisper:
    cfg: WhisperCfg
    conv1_w: torch.Tensor   # [d, 3*128]  (tap-major, mel padded to 128)
    conv1_b: torch.Tensor
    conv2_w: torch.Tensor   # [d, 3*d]
    conv2_b: torch.Tensor
    pos: torch.Tensor       # f32 [1500, d]
    layers: List[EncLayer]
    lnf_g: torch.Tensor
    lnf_b: torch.Tensor


def _pack_enc_layer(sd: SD, lp: str, d: int, device, consume: bool, k_bias: bool, ln1: str, ln2: str) -> EncLayer:
    q = _take(sd, lp + "self_attn.q_proj.weight", consume)
    k = _take(sd, lp + "self_attn.k_proj.weight", consume)
    v = _take(sd, lp + "self_attn.v_proj.weight", consume)
    wqkv = _bf(torch.cat([q, k, v], 0), device)
    bq = _take(sd, lp + "self_attn.q_proj.bias", consume)
    bk = _take(sd, lp + "self_attn.k_proj.bias", consume) if k_bias else torch.zeros_like(bq)
    bv = _take(sd, lp + "self_attn.v_proj.bias", consume)
    return EncLayer(
        ln1_g=_f32(_take(sd, lp + ln1 + ".weight", consume), device), ln1_b=_f32(_take(sd, lp + ln1 + ".bias", consume), device),
        wqkv=wqkv, bqkv=_f32(torch.cat([bq.float(), bk.float(), bv.float()]), device),
        wo=_bf(_take(sd, lp + "self_attn.out_proj.weight", consume), device),
        bo=_f32(_take(sd, lp + "self_attn.out_proj.bias", consume), device),
        ln2_g=_f32(_take(sd, lp + ln2 + ".weight", consume), device), ln2_b=_f32(_take(sd, lp + ln2 + ".bias", consume), device),
        w1=_bf(_take(sd, lp + "fc1.weight", consume), device), b1=_f32(_take(sd, lp + "fc1.bias", consume), device),
        w2=_bf(_take(sd, lp + "fc2.weight", consume), device), b2=_f32(_take(sd, lp + "fc2.bias", consume), device))


def pack_whisper(sd: SD, cfg: WhisperCfg, device, prefix: str = "speech_encoder.", consume: bool = False) -> PackedWhisper:
    d = cfg.d_model
    assert d % 64 == 0 and cfg.ffn % 64 == 0 and d // cfg.n_heads in (64, 128), "HIP path needs d%64==0, head_dim 64/128"
    assert cfg.n_mels <= 128
    c1 = _take(sd, prefix + "conv1.weight", consume).float()             # [d, n_mels, 3]
    w1 = torch.zeros(d, 3, 128, dtype=torch.float32, device=c1.device)
    w1[:, :, :cfg.n_mels] = c1.permute(0, 2, 1)
    c2 = _take(sd, prefix + "conv2.weight", consume)                     # [d, d, 3]
    layers = [_pack_enc_layer(sd, f"{prefix}layers.{i}.", d, device, consume, False, "self_attn_layer_norm",
                              "final_layer_norm") for i in range(cfg.n_layers)]
    return PackedWhisper(
        cfg=cfg, conv1_w=_bf(w1.reshape(d, 384), device), conv1_b=_f32(_take(sd, prefix + "conv1.bias", consume), device),
        conv2_w=_bf(c2.permute(0, 2, 1).reshape(d, 3 * d), device),
        conv2_b=_f32(_take(sd, prefix + "conv2.bias", consume), device),
        pos=_f32(_take(sd, prefix + "embed_positions.weight", consume), device), layers=layers,
        lnf_g=_f32(_take(sd, prefix + "layer_norm.weight", consume), device),
        lnf_b=_f32(_take(sd, prefix + "layer_norm.bias", consume), device))


# ------------------------------------------------------------------------------------------------
@dataclass
class PackedBeats:
    cfg: BeatsCfg
    patch_w: torch.Tensor    # [embed, 256]
    ln0_g: torch.Tensor
    ln0_b: torch.Tensor
    proj_w: torch.Tensor     # [d, embed]
    proj_b: torch.Tensor
    posconv_w: torch.Tensor  # [groups, d/groups, conv_pos * d/groups]  (tap-major)
    posconv_b: torch.Tensor
    enc_ln_g: torch.Tensor
    enc_ln_b: torch.Tensor
    rel_table: torch.Tensor  # f32 [heads, 2*span-1]
    rel_span: int
    layers: List[EncLayer]


def beats_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional T5/WavLM-style bucket of (key - query) offsets (BEATs backbone.py, restated)."""
    import math
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.clamp(large, max=nb - 1)
    return ret + torch.where(n < max_exact, n, large)


def pack_beats(sd: SD, cfg: BeatsCfg, device, prefix: str = "beats.", span: int = 1504, consume: bool = False) -> PackedBeats:
    d, G = cfg.d_model, cfg.conv_groups
    cpg = d // G
    assert d % 64 == 0 and cfg.embed % 64 == 0 and d // cfg.n_heads == 64 and cpg % 8 == 0 and cfg.conv_pos == 128
    pw = _take(sd, prefix + "patch_embedding.weight", consume)            # [embed, 1, 16, 16]
    g = _take(sd, prefix + "encoder.pos_conv.0.weight_g", consume).float()
    v = _take(sd, prefix + "encoder.pos_conv.0.weight_v", consume).float()
    w = v * (g / v.norm(dim=(0, 1), keepdim=True))                        # weight_norm(dim=2): [d, cpg, 128]
    wg = w.view(G, cpg, cpg, cfg.conv_pos).permute(0, 1, 3, 2).reshape(G, cpg, cfg.conv_pos * cpg)
    emb = _take(sd, prefix + "encoder.layers.0.self_attn.relative_attention_bias.weight", consume).float()
    rel = torch.arange(-(span - 1), span)
    table = emb.cpu()[beats_bucket(rel, cfg.num_buckets, cfg.max_distance)].t().contiguous()   # [heads, 2*span-1]
    layers = []
    for i in range(cfg.n_layers):
        lp = f"{prefix}encoder.layers.{i}."
        L = _pack_enc_layer(sd, lp, d, device, consume, True, "self_attn_layer_norm", "final_layer_norm")
        L.extra = dict(grep_w=_f32(_take(sd, lp + "self_attn.grep_linear.weight", consume), device),
                       grep_b=_f32(_take(sd, lp + "self_attn.grep_linear.bias", consume), device),
                       grep_a=_f32(_take(sd, lp + "self_attn.grep_a", consume).reshape(-1), device))
        layers.append(L)
    return PackedBeats(
        cfg=cfg, patch_w=_bf(pw.reshape(cfg.embed, 256), device),
        ln0_g=_f32(_take(sd, prefix + "layer_norm.weight", consume), device),
        ln0_b=_f32(_take(sd, prefix + "layer_norm.bias", consume), device),
        proj_w=_bf(_take(sd, prefix + "post_extract_proj.weight", consume), device),
        proj_b=_f32(_take(sd, prefix + "post_extract_proj.bias", consume), device),
        posconv_w=_bf(wg, device), posconv_b=_f32(_take(sd, prefix + "encoder.pos_conv.0.bias", consume), device),
        enc_ln_g=_f32(_take(sd, prefix + "encoder.layer_norm.weight", consume), device),
        enc_ln_b=_f32(_take(sd, prefix + "encoder.layer_norm.bias", consume), device),
        rel_table=_f32(table, device), rel_span=span, layers=layers)


# ------------------------------------------------------------------------------------------------
@dataclass
class QFLayer:
    sa_wv: torch.Tensor; sa_bv: torch.Tensor; sa_wo: torch.Tensor; sa_bo: torch.Tensor
    sa_ln_g: torch.Tensor; sa_ln_b: torch.Tensor
    ca_wq: torch.Tensor; ca_bq: torch.Tensor; ca_wkv: torch.Tensor; ca_bkv: torch.Tensor
    ca_wo: torch.Tensor; ca_bo: torch.Tensor; ca_ln_g: torch.Tensor; ca_ln_b: torch.Tensor
    w1: torch.Tensor; b1: torch.Tensor; w2: torch.Tensor; b2: torch.Tensor
    ff_ln_g: torch.Tensor; ff_ln_b: torch.Tensor


@dataclass
class PackedQFormer:
    cfg: QFormerCfg
    ln_speech_g: torch.Tensor; ln_speech_b: torch.Tensor
    ln_audio_g: Optional[torch.Tensor]; ln_audio_b: Optional[torch.Tensor]
    query: torch.Tensor      # f32 [1, hidden]
    emb_ln_g: torch.Tensor; emb_ln_b: torch.Tensor
    layers: List[QFLayer]
    proj_w: torch.Tensor; proj_b: torch.Tensor


def pack_qformer(sd: SD, cfg: QFormerCfg, device, consume: bool = False) -> PackedQFormer:
    assert cfg.n_query == 1, "HIP path implements the window-level Q-Former with one query token per window"
    assert cfg.hidden % 64 == 0 and cfg.hidden // cfg.n_heads == 64 and cfg.enc_width % 64 == 0 and cfg.ffn % 64 == 0
    p = "speech_Qformer.bert."
    t = lambda k: _take(sd, k, consume)
    layers = []
    for i in range(cfg.n_layers):
        lp = f"{p}encoder.layer.{i}."
        # with ONE query token the self-attention softmax is identically 1: only value/out projections matter
        for dead in ("attention.self.query.weight", "attention.self.query.bias", "attention.self.key.weight",
                     "attention.self.key.bias"):
            if consume:
                sd.pop(lp + dead, None)
        layers.append(QFLayer(
            sa_wv=_bf(t(lp + "attention.self.value.weight"), device), sa_bv=_f32(t(lp + "attention.self.value.bias"), device),
            sa_wo=_bf(t(lp + "attention.output.dense.weight"), device), sa_bo=_f32(t(lp + "attention.output.dense.bias"), device),
            sa_ln_g=_f32(t(lp + "attention.output.LayerNorm.weight"), device),
            sa_ln_b=_f32(t(lp + "attention.output.LayerNorm.bias"), device),
            ca_wq=_bf(t(lp + "crossattention.self.query.weight"), device),
            ca_bq=_f32(t(lp + "crossattention.self.query.bias"), device),
            ca_wkv=_bf(torch.cat([t(lp + "crossattention.self.key.weight"), t(lp + "crossattention.self.value.weight")], 0), device),
            ca_bkv=_f32(torch.cat([t(lp + "crossattention.self.key.bias").float(), t(lp + "crossattention.self.value.bias").float()]), device),
            ca_wo=_bf(t(lp + "crossattention.output.dense.weight"), device),
            ca_bo=_f32(t(lp + "crossattention.output.dense.bias"), device),
            ca_ln_g=_f32(t(lp + "crossattention.output.LayerNorm.weight"), device),
            ca_ln_b=_f32(t(lp + "crossattention.output.LayerNorm.bias"), device),
            w1=_bf(t(lp + "intermediate_query.dense.weight"), device), b1=_f32(t(lp + "intermediate_query.dense.bias"), device),
            w2=_bf(t(lp + "output_query.dense.weight"), device), b2=_f32(t(lp + "output_query.dense.bias"), device),
            ff_ln_g=_f32(t(lp + "output_query.LayerNorm.weight"), device),
            ff_ln_b=_f32(t(lp + "output_query.LayerNorm.bias"), device)))
    has_audio = "ln_audio.weight" in sd
    return PackedQFormer(
        cfg=cfg, ln_speech_g=_f32(t("ln_speech.weight"), device), ln_speech_b=_f32(t("ln_speech.bias"), device),
        ln_audio_g=_f32(t("ln_audio.weight"), device) if has_audio else None,
        ln_audio_b=_f32(t("ln_audio.bias"), device) if has_audio else None,
        query=_f32(t("speech_query_tokens").reshape(1, cfg.hidden), device),
        emb_ln_g=_f32(t(p + "embeddings.LayerNorm.weight"), device), emb_ln_b=_f32(t(p + "embeddings.LayerNorm.bias"), device),
        layers=layers, proj_w=_bf(t("speech_llama_proj.weight"), device), proj_b=_f32(t("speech_llama_proj.bias"), device))


# ------------------------------------------------------------------------------------------------
@dataclass
class LlamaLayer:
    rms1: torch.Tensor
    bqkv: Optional[torch.Tensor]  # f32 [3h] (Qwen2) or None
    wqkv: torch.Tensor           # [3h, K_aug]
    lora_a: Optional[torch.Tensor]  # [2r, h], pre-multiplied by alpha/r
    wo: torch.Tensor
    rms2: torch.Tensor
    wgu: torch.Tensor            # [2*ffn, h] interleaved in blocks of 16
    wdown: torch.Tensor
    decode_packed: Optional[tuple] = None   # (wqkv, wo, wgu, wdown) in the M <= 128 decode tile's layout, made on first use


@dataclass
class PackedLlama:
    cfg: LlamaCfg
    embed: torch.Tensor
    layers: List[LlamaLayer]
    norm: torch.Tensor
    lm_head: torch.Tensor
    k_aug: int
    rope_cos: torch.Tensor       # f32 [max_pos, head_dim/2]
    rope_sin: torch.Tensor


def pack_llama(sd: SD, cfg: LlamaCfg, device, prefix: str = "llama_model.", consume: bool = False) -> PackedLlama:
    """HF causal-LM names under `prefix`: model.embed_tokens, model.layers.{i}.*, model.norm, lm_head (Llama and Qwen2 alike)."""
    h, I, r = cfg.hidden, cfg.ffn, cfg.lora_rank
    assert h % 64 == 0 and I % 64 == 0 and cfg.head_dim in (64, 128)
    p = prefix + "model."
    k_aug = h + (LORA_PAD if r else 0)
    assert len(cfg.lora_targets) * r <= LORA_PAD and all(t in ("q_proj", "k_proj", "v_proj") for t in cfg.lora_targets)
    layers = []
    for i in range(cfg.n_layers):
        lp = f"{p}layers.{i}."
        wqkv = torch.zeros(3 * h, k_aug, dtype=torch.bfloat16, device=device)
        for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
            wqkv[j * h:(j + 1) * h, :h] = _take(sd, lp + f"self_attn.{n}.weight", consume).to(device=device, dtype=torch.bfloat16)
        lora_a = None
        if r:   # K-augmentation: target t (one of q/k/v) gets its B in column block [h + ti*r, h + (ti+1)*r) of its own rows
            a_rows = []
            for ti, tgt in enumerate(cfg.lora_targets):
                j = ("q_proj", "k_proj", "v_proj").index(tgt)
                a_rows.append(_take(sd, lp + f"self_attn.{tgt}.lora_A.weight", consume))
                wqkv[j * h:(j + 1) * h, h + ti * r:h + (ti + 1) * r] = \
                    _take(sd, lp + f"self_attn.{tgt}.lora_B.weight", consume).to(device=device, dtype=torch.bfloat16)
            # the LoRA scale (alpha/r) is folded into A here so the down-projection can run as a plain GEMM
            lora_a = _bf(torch.cat(a_rows, 0).float() * cfg.lora_scale, device)
        bqkv = None
        if cfg.qkv_bias:
            bqkv = _f32(torch.cat([_take(sd, lp + f"self_attn.{n}.bias", consume).float() for n in ("q_proj", "k_proj", "v_proj")]), device)
        g = _take(sd, lp + "mlp.gate_proj.weight", consume).to(device=device, dtype=torch.bfloat16)
        u = _take(sd, lp + "mlp.up_proj.weight", consume).to(device=device, dtype=torch.bfloat16)
        wgu = torch.stack([g.view(I // 16, 16, h), u.view(I // 16, 16, h)], dim=1).reshape(2 * I, h).contiguous()
        del g, u
        layers.append(LlamaLayer(
            rms1=_f32(_take(sd, lp + "input_layernorm.weight", consume), device), bqkv=bqkv, wqkv=wqkv, lora_a=lora_a,
            wo=_bf(_take(sd, lp + "self_attn.o_proj.weight", consume), device),
            rms2=_f32(_take(sd, lp + "post_attention_layernorm.weight", consume), device), wgu=wgu,
            wdown=_bf(_take(sd, lp + "mlp.down_proj.weight", consume), device)))
    D = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32) / D))
    ang = torch.arange(cfg.max_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return PackedLlama(
        cfg=cfg, embed=_bf(_take(sd, p + "embed_tokens.weight", consume), device), layers=layers,
        norm=_f32(_take(sd, p + "norm.weight", consume), device),
        lm_head=_bf(_take(sd, prefix + "lm_head.weight", consume), device), k_aug=k_aug,
        rope_cos=_f32(ang.cos(), device), rope_sin=_f32(ang.sin(), device))

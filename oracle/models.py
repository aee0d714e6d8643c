"""ORACLE (test infrastructure only — never imported by the product path).

Plain PyTorch-CPU restatement of the arithmetic of the ICL forward/generate hot path, written from
the math (no `transformers` import here).  The reference keeps NONE of this arithmetic in-tree
(SURVEY.md §0.2): it reaches it through

* ``models/custom_salmon.py:550-554``  -> ``SALMONN.encode_speech`` (external SALMONN package):
  Whisper encoder (K2,K3), BEATs (K4,K5), ln_speech/ln_audio + window unfold (K6), the 2-layer
  window-level Q-Former (K7) and ``speech_llama_proj`` (K8);
* ``models/custom_salmon.py:630-636``  -> ``llama_model(inputs_embeds=…, labels=…)`` (K10, K12);
* ``models/custom_salmon.py:704-720``  -> ``llama_model.generate(...)`` greedy (K10, K11).

Third-party sources restated (none vendored in /root/reference, none version-pinned by it):
``transformers`` Whisper / Llama modelling code (installed here: 5.15.0 — used by
tests/golden/make_golden.py to pin this file through golden vectors), peft LoRA
(``W x + (alpha/r) B A x``; not installed), bytedance/SALMONN ``Qformer.py`` (BLIP-2 BERT with
cross-attention; pinned architecturally against the installed ``Blip2QFormerModel``) and
microsoft/unilm BEATs (``BEATs.py`` / ``backbone.py``; no copy in this container: **parity
unpinned vs upstream**).

All functions take a flat ``dict[str, Tensor]`` of weights under the checkpoint key names of the
reference stack (HF / SALMONN ``state_dict`` names) and run in fp32.  ``rnd`` is an optional
rounding hook applied at exactly the points where the HIP path stores bf16 (GEMM operands,
attention outputs); with ``rnd=bf16_round`` the oracle reproduces the HIP path's rounding points
and is compared with a tight tolerance, with ``rnd=None`` it is the reference's fp32 CPU behaviour.
"""
from __future__ import annotations

import math

import numpy as np
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def bf16_round(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def bf16_round_activations(sd: "SD") -> Callable:
    """``bf16_round`` for ACTIVATIONS only: tensors that are entries of ``sd`` pass through untouched.  Valid when the weights
    are already bf16-representable (checkpoints generated or stored in bf16) — the rounding of a weight is then the identity
    and skipping it saves a full pass over the parameters per call (7B: ~10 s of host time per decode step otherwise)."""
    ptrs = {v.data_ptr() for v in sd.values() if v.dtype == torch.float32}

    def rnd(t: Tensor) -> Tensor:
        return t if t.data_ptr() in ptrs else bf16_round(t)
    return rnd


def _id(t: Tensor) -> Tensor:
    return t


def _lin(x: Tensor, sd: SD, name: str, rnd) -> Tensor:
    """y = rnd(x) @ rnd(W)^T + b  (bias stays fp32, as in the HIP epilogue)."""
    w = rnd(sd[name + ".weight"].float())
    y = rnd(x) @ w.t()
    b = sd.get(name + ".bias")
    return y if b is None else y + b.float()


def _ln(x: Tensor, sd: SD, name: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"].float(), sd[name + ".bias"].float(), eps)


def _mha(q: Tensor, k: Tensor, v: Tensor, n_heads: int, scale: float, mask: Optional[Tensor] = None,
         bias: Optional[Tensor] = None) -> Tensor:
    """q [B,Tq,C], k/v [B,Tk,C] -> [B,Tq,C]; mask True = masked, broadcastable to [B,H,Tq,Tk]."""
    B, Tq, C = q.shape
    Tk = k.shape[1]
    D = C // n_heads
    qh = q.view(B, Tq, n_heads, D).transpose(1, 2)
    kh = k.view(B, Tk, n_heads, D).transpose(1, 2)
    vh = v.view(B, Tk, n_heads, D).transpose(1, 2)
    sc = (qh @ kh.transpose(-1, -2)) * scale
    if bias is not None:
        sc = sc + bias
    if mask is not None:
        sc = sc.masked_fill(mask, float("-inf"))
    o = torch.softmax(sc, dim=-1) @ vh
    return o.transpose(1, 2).reshape(B, Tq, C)


# ---------------------------------------------------------------------------------------------
# K2 + K3: Whisper encoder (HF WhisperEncoder semantics)
# ---------------------------------------------------------------------------------------------
def whisper_encoder(sd: SD, spec: Tensor, n_heads: int, prefix: str = "", rnd: Optional[Callable] = None,
                    return_layers: bool = False, key_lens: Optional[List[int]] = None, final_ln: bool = True):
    """spec f32 [B, n_mel, 3000] -> [B, 1500, d].  Keys: conv1/conv2, embed_positions.weight,
    layers.{i}.{self_attn.{q,k,v,out}_proj, self_attn_layer_norm, fc1, fc2, final_layer_norm}, layer_norm."""
    rnd = rnd or _id
    p = prefix
    w1, b1 = rnd(sd[p + "conv1.weight"].float()), sd[p + "conv1.bias"].float()
    w2, b2 = rnd(sd[p + "conv2.weight"].float()), sd[p + "conv2.bias"].float()
    x = rnd(F.gelu(F.conv1d(rnd(spec.float()), w1, b1, padding=1)))
    x = F.gelu(F.conv1d(x, w2, b2, stride=2, padding=1))
    h = x.permute(0, 2, 1) + sd[p + "embed_positions.weight"].float()[None]
    d = h.shape[-1]
    scale = (d // n_heads) ** -0.5
    n_layers = 1 + max(int(k[len(p):].split(".")[1]) for k in sd if k.startswith(p + "layers."))
    layers = []
    kmask = None
    if key_lens is not None:   # Qwen2-Audio: encoder frames past the audio are masked as keys
        kmask = (torch.arange(h.shape[1])[None, :] >= torch.tensor(key_lens)[:, None])[:, None, None, :]
    for i in range(n_layers):
        lp = f"{p}layers.{i}."
        xn = _ln(h, sd, lp + "self_attn_layer_norm", 1e-5)
        q = rnd(_lin(xn, sd, lp + "self_attn.q_proj", rnd))
        k = rnd(_lin(xn, sd, lp + "self_attn.k_proj", rnd))
        v = rnd(_lin(xn, sd, lp + "self_attn.v_proj", rnd))
        a = rnd(_mha(q, k, v, n_heads, scale, mask=kmask))
        h = h + _lin(a, sd, lp + "self_attn.out_proj", rnd)
        xn = _ln(h, sd, lp + "final_layer_norm", 1e-5)
        f = rnd(F.gelu(_lin(xn, sd, lp + "fc1", rnd)))
        h = h + _lin(f, sd, lp + "fc2", rnd)
        if return_layers:
            layers.append(h.clone())
    if not final_ln:
        return h
    out = _ln(h, sd, p + "layer_norm", 1e-5)
    return (out, layers) if return_layers else out


# ---------------------------------------------------------------------------------------------
# K13: Qwen2-Audio tower + projector (HF Qwen2AudioEncoder / Qwen2AudioMultiModalProjector semantics)
# ---------------------------------------------------------------------------------------------
def qwen_audio_lengths(mel_len: int) -> Tuple[int, int]:
    feat = (mel_len - 1) // 2 + 1
    return feat, (feat - 2) // 2 + 1


def qwen_audio_features(sd: SD, spec: Tensor, mel_lens: List[int], n_heads: int, rnd: Optional[Callable] = None):
    """spec f32 [n, 128, 3000] -> (projected features [n, 750, H_llm], valid rows per audio).  Keys: ``audio_tower.*``,
    ``multi_modal_projector.linear.*``.  Reference call: models/custom_qwen.py:186-195 -> HF Qwen2AudioModel.forward."""
    rnd = rnd or _id
    lens = [qwen_audio_lengths(int(m)) for m in mel_lens]
    h = whisper_encoder(sd, spec, n_heads, prefix="audio_tower.", rnd=rnd, key_lens=[f for f, _ in lens], final_ln=False)
    pooled = F.avg_pool1d(h.transpose(1, 2), 2, 2).transpose(1, 2)
    x = _ln(pooled, sd, "audio_tower.layer_norm", 1e-5)
    return _lin(x, sd, "multi_modal_projector.linear", rnd), [o for _, o in lens]


# ---------------------------------------------------------------------------------------------
# K4 + K5: BEATs (microsoft/unilm BEATs.py + backbone.py, iter3+ config: post-LN, deep-norm,
# gated relative position bias)
# ---------------------------------------------------------------------------------------------
def beats_relative_buckets(rel: Tensor, num_buckets: int = 320, max_distance: int = 800) -> Tensor:
    """T5-style bidirectional bucketing of (memory - context) positions."""
    nb = num_buckets // 2
    ret = (rel > 0).long() * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp_min(1) / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


def beats_position_bias_table(sd: SD, span: int, prefix: str = "", num_buckets: int = 320,
                              max_distance: int = 800) -> Tensor:
    """[n_heads, 2*span-1]: bias for relative offsets -(span-1) .. span-1 (shared by all layers)."""
    rel = torch.arange(-(span - 1), span)
    emb = sd[prefix + "encoder.layers.0.self_attn.relative_attention_bias.weight"].float()  # [buckets, heads]
    return emb[beats_relative_buckets(rel, num_buckets, max_distance)].t().contiguous()


def beats_encoder(sd: SD, wav: Tensor, wav_lens: List[int], prefix: str = "", n_heads: int = 12,
                  num_buckets: int = 320, max_distance: int = 800, rnd: Optional[Callable] = None,
                  fbank: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """wav f32 [B, Lmax] (zero padded), wav_lens -> (features [B, T, 768], padding_mask [B, T] bool).

    Mirrors BEATs.extract_features(source, padding_mask) with padding_mask[b, t] = t >= wav_lens[b]."""
    from .audio_frontend import kaldi_fbank, kaldi_num_frames
    rnd = rnd or _id
    p = prefix
    B, Lmax = wav.shape
    if fbank is None:
        nf = kaldi_num_frames(Lmax)
        fbank = torch.stack([torch.from_numpy(kaldi_fbank(wav[b].numpy())) for b in range(B)])  # [B, nf, 128]
    nf = fbank.shape[1]
    pad_wav = torch.arange(Lmax)[None, :] >= torch.tensor(wav_lens)[:, None]

    def fwd_mask(feat_len: int, mask: Tensor) -> Tensor:   # BEATs.forward_padding_mask
        extra = mask.size(1) % feat_len
        if extra > 0:
            mask = mask[:, :-extra]
        return mask.view(mask.size(0), feat_len, -1).all(-1)

    pm = fwd_mask(nf, pad_wav)
    x = F.conv2d(rnd(fbank.float())[:, None], rnd(sd[p + "patch_embedding.weight"].float()), stride=16)
    x = x.reshape(B, x.shape[1], -1).transpose(1, 2)            # [B, T, 512], T = (nf//16)*8
    T = x.shape[1]
    x = _ln(x, sd, p + "layer_norm", 1e-5)
    pm = fwd_mask(T, pm)
    x = _lin(x, sd, p + "post_extract_proj", rnd)
    # ---- TransformerEncoder ----
    x = x.masked_fill(pm[..., None], 0.0)
    g, v = sd[p + "encoder.pos_conv.0.weight_g"].float(), sd[p + "encoder.pos_conv.0.weight_v"].float()
    w = v * (g / v.norm(dim=(0, 1), keepdim=True))             # weight_norm(dim=2)
    xc = F.conv1d(rnd(x).transpose(1, 2), rnd(w), sd[p + "encoder.pos_conv.0.bias"].float(), padding=64, groups=16)
    xc = F.gelu(xc[:, :, :-1]).transpose(1, 2)                  # SamePad(128) drops the last frame
    x = _ln(x + xc, sd, p + "encoder.layer_norm", 1e-5)
    C = x.shape[-1]
    D = C // n_heads
    n_layers = 1 + max(int(k[len(p):].split(".")[2]) for k in sd if k.startswith(p + "encoder.layers."))
    alpha = (2.0 * n_layers) ** 0.25                            # deep_norm_alpha
    table = beats_position_bias_table(sd, T, p, num_buckets, max_distance)          # [H, 2T-1]
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1
    pos_bias = table[:, idx]                                                         # [H, T, T]
    kmask = pm[:, None, None, :]
    for i in range(n_layers):
        lp = f"{p}encoder.layers.{i}."
        q = rnd(_lin(x, sd, lp + "self_attn.q_proj", rnd))
        k = rnd(_lin(x, sd, lp + "self_attn.k_proj", rnd))
        vv = rnd(_lin(x, sd, lp + "self_attn.v_proj", rnd))
        qh = q.view(B, T, n_heads, D).transpose(1, 2)                                 # unscaled, with bias
        gl = qh @ sd[lp + "self_attn.grep_linear.weight"].float().t() + sd[lp + "self_attn.grep_linear.bias"].float()
        gab = torch.sigmoid(gl.view(B, n_heads, T, 2, 4).sum(-1))
        ga, gb = gab[..., 0:1], gab[..., 1:2]
        gate = ga * (gb * sd[lp + "self_attn.grep_a"].float().view(1, n_heads, 1, 1) - 1.0) + 2.0   # [B,H,T,1]
        a = rnd(_mha(q, k, vv, n_heads, D ** -0.5, mask=kmask, bias=gate * pos_bias[None]))
        x = _ln(x * alpha + _lin(a, sd, lp + "self_attn.out_proj", rnd), sd, lp + "self_attn_layer_norm", 1e-5)
        f = rnd(F.gelu(_lin(x, sd, lp + "fc1", rnd)))
        x = _ln(x * alpha + _lin(f, sd, lp + "fc2", rnd), sd, lp + "final_layer_norm", 1e-5)
    return x, pm


# ---------------------------------------------------------------------------------------------
# K6 + K7 + K8: ln_speech / ln_audio, concat, window unfold, Q-Former, projector
# ---------------------------------------------------------------------------------------------
def qformer(sd: SD, query: Tensor, enc: Tensor, n_heads: int = 12, prefix: str = "speech_Qformer.bert.",
            rnd: Optional[Callable] = None, eps: float = 1e-12) -> Tensor:
    """query [W, nq, 768], enc [W, win, C_enc] -> [W, nq, 768] (BERT post-LN, cross-attention in every layer,
    query-only FFN: SALMONN Qformer.py with cross_attention_freq=1)."""
    rnd = rnd or _id
    p = prefix
    h = _ln(query.float(), sd, p + "embeddings.LayerNorm", eps)
    C = h.shape[-1]
    scale = (C // n_heads) ** -0.5
    n_layers = 1 + max(int(k[len(p):].split(".")[2]) for k in sd if k.startswith(p + "encoder.layer."))
    for i in range(n_layers):
        lp = f"{p}encoder.layer.{i}."
        q = rnd(_lin(h, sd, lp + "attention.self.query", rnd))
        k = rnd(_lin(h, sd, lp + "attention.self.key", rnd))
        v = rnd(_lin(h, sd, lp + "attention.self.value", rnd))
        a = rnd(_mha(q, k, v, n_heads, scale))
        h = _ln(_lin(a, sd, lp + "attention.output.dense", rnd) + h, sd, lp + "attention.output.LayerNorm", eps)
        q = rnd(_lin(h, sd, lp + "crossattention.self.query", rnd))
        k = rnd(_lin(enc, sd, lp + "crossattention.self.key", rnd))
        v = rnd(_lin(enc, sd, lp + "crossattention.self.value", rnd))
        a = rnd(_mha(q, k, v, n_heads, scale))
        h = _ln(_lin(a, sd, lp + "crossattention.output.dense", rnd) + h, sd, lp + "crossattention.output.LayerNorm", eps)
        f = rnd(F.gelu(_lin(h, sd, lp + "intermediate_query.dense", rnd)))
        h = _ln(_lin(f, sd, lp + "output_query.dense", rnd) + h, sd, lp + "output_query.LayerNorm", eps)
    return h


def salmonn_window_count(T: int = 1500, second_per_window: float = 0.333333, second_stride: float = 0.333333):
    kernel = round(1500 * second_per_window / 30.0)
    stride = round(1500 * second_stride / 30.0)
    return kernel, stride, (T - kernel) // stride + 1


def salmonn_fuse_qformer(sd: SD, speech: Tensor, audio: Optional[Tensor], rnd: Optional[Callable] = None,
                         second_per_window: float = 0.333333, second_stride: float = 0.333333,
                         qformer_heads: int = 12) -> Tensor:
    """SALMONN._encode_auditory_feature: speech [B,1500,1280] (Whisper out), audio [B,Ta,768] (BEATs out or None)
    -> [B, n_windows, H_llm]."""
    rnd = rnd or _id
    s = _ln(speech, sd, "ln_speech", 1e-5)
    if audio is not None:
        a = _ln(audio, sd, "ln_audio", 1e-5)
        if a.shape[1] < s.shape[1]:
            a = F.pad(a, (0, 0, 0, s.shape[1] - a.shape[1]))
        elif a.shape[1] > s.shape[1]:
            s = F.pad(s, (0, 0, 0, a.shape[1] - s.shape[1]))
        s = torch.cat([s, a], dim=-1)
    s = rnd(s)
    B, T, C = s.shape
    kernel, stride, L = salmonn_window_count(T, second_per_window, second_stride)
    tr = s.transpose(1, 2).unsqueeze(2)
    ov = F.unfold(tr, kernel_size=(1, kernel), dilation=1, padding=0, stride=(1, stride))
    ov = ov.view(B, -1, kernel, L).permute(0, 3, 2, 1).reshape(-1, kernel, C)
    qt = sd["speech_query_tokens"].float().expand(ov.shape[0], -1, -1)
    qo = qformer(sd, qt, ov, n_heads=qformer_heads, rnd=rnd)
    out = _lin(qo, sd, "speech_llama_proj", rnd)
    return out.view(B, -1, out.shape[-1]).contiguous()


def salmonn_encode_speech(sd: SD, spec: Tensor, wav: Optional[Tensor], wav_lens: Optional[List[int]],
                          whisper_heads: int, use_beats: bool = True, rnd: Optional[Callable] = None,
                          beats_cfg: Optional[dict] = None, qformer_heads: int = 12) -> Tensor:
    """SALMONN.encode_speech(spectrogram, raw_wav, audio_padding_mask) -> [B, 88, H_llm]."""
    speech = whisper_encoder(sd, spec, whisper_heads, prefix="speech_encoder.", rnd=rnd)
    audio = None
    if use_beats and wav is not None:
        audio, _ = beats_encoder(sd, wav, wav_lens, prefix="beats.", rnd=rnd, **(beats_cfg or {}))
    return salmonn_fuse_qformer(sd, speech, audio, rnd=rnd, qformer_heads=qformer_heads)


# ---------------------------------------------------------------------------------------------
# K10 + K11 + K12: Llama decoder (HF LlamaForCausalLM semantics) with optional un-merged LoRA
# ---------------------------------------------------------------------------------------------
class LlamaOracle:
    """Weights under HF names (``model.layers.{i}...``, ``model.norm.weight``, ``lm_head.weight``,
    ``model.embed_tokens.weight``); LoRA as ``model.layers.{i}.self_attn.{q,v}_proj.lora_{A,B}.weight``."""

    def __init__(self, sd: SD, n_heads: int, rms_eps: float = 1e-5, rope_theta: float = 10000.0,
                 lora_scale: float = 0.0, rnd: Optional[Callable] = None, n_kv_heads: Optional[int] = None):
        self.sd, self.H, self.eps, self.theta = sd, n_heads, rms_eps, rope_theta
        self.lora_scale = lora_scale
        self.rnd = rnd or _id
        self.n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("model.layers."))
        self.hidden = sd["model.norm.weight"].shape[0]
        self.D = self.hidden // n_heads
        self.Hkv = n_kv_heads or n_heads

    def embed(self, ids: Tensor) -> Tensor:
        # the HIP path keeps the embedding table in bf16 (rnd = bf16_round reproduces that)
        return self.rnd(self.sd["model.embed_tokens.weight"].float()[ids])

    def _rms(self, x: Tensor, name: str) -> Tensor:
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.sd[name].float()

    def _proj(self, x: Tensor, name: str) -> Tensor:
        y = _lin(x, self.sd, name, self.rnd)
        a = self.sd.get(name + ".lora_A.weight")
        if a is not None and self.lora_scale != 0.0:
            t = self.rnd((self.rnd(x) @ self.rnd(a.float()).t()) * self.lora_scale)
            y = y + t @ self.rnd(self.sd[name + ".lora_B.weight"].float()).t()
        return y

    def _rope(self, x: Tensor, pos: Tensor) -> Tensor:  # x [B,T,H,D], pos [B,T]
        half = self.D // 2
        inv = 1.0 / (self.theta ** (torch.arange(0, self.D, 2).float() / self.D))
        ang = pos.float()[..., None] * inv
        c, s = ang.cos()[:, :, None, :], ang.sin()[:, :, None, :]
        x1, x2 = x[..., :half], x[..., half:]
        return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1)

    def forward_hidden(self, h: Tensor, pos: Tensor, cache: Optional[list] = None, attn_mask: Optional[Tensor] = None):
        """h [B,T,C] f32, pos [B,T] -> final hidden (pre-norm) [B,T,C].  cache: list of (k,v) per layer (appended).
        attn_mask: optional bool [B, Tk] True = key visible (left/right padding)."""
        rnd = self.rnd
        B, T, C = h.shape
        for i in range(self.n_layers):
            lp = f"model.layers.{i}."
            xn = self._rms(h, lp + "input_layernorm.weight")
            q = self._proj(xn, lp + "self_attn.q_proj").view(B, T, self.H, self.D)
            k = self._proj(xn, lp + "self_attn.k_proj").view(B, T, self.Hkv, self.D)
            v = rnd(self._proj(xn, lp + "self_attn.v_proj")).view(B, T, self.Hkv, self.D)
            q, k = rnd(self._rope(rnd(q), pos)), rnd(self._rope(rnd(k), pos))
            if cache is not None:
                if len(cache) <= i:
                    cache.append((k, v))
                else:
                    cache[i] = (torch.cat([cache[i][0], k], 1), torch.cat([cache[i][1], v], 1))
                k, v = cache[i]
            Tk = k.shape[1]
            if self.Hkv != self.H:
                rep = self.H // self.Hkv
                k, v = k.repeat_interleave(rep, dim=2), v.repeat_interleave(rep, dim=2)
            mask = torch.arange(Tk)[None, :] > (torch.arange(T)[:, None] + (Tk - T))      # causal
            mask = mask[None, None]
            if attn_mask is not None:
                mask = mask | ~attn_mask[:, None, None, :].bool()
            a = rnd(_mha(q.reshape(B, T, C), k.reshape(B, Tk, C), v.reshape(B, Tk, C), self.H, self.D ** -0.5, mask=mask))
            h = h + self._proj(a, lp + "self_attn.o_proj")
            xn = self._rms(h, lp + "post_attention_layernorm.weight")
            g = _lin(xn, self.sd, lp + "mlp.gate_proj", rnd)
            u = _lin(xn, self.sd, lp + "mlp.up_proj", rnd)
            h = h + _lin(rnd(F.silu(g) * u), self.sd, lp + "mlp.down_proj", rnd)
        return h

    def logits(self, h: Tensor) -> Tensor:
        return _lin(self._rms(h, "model.norm.weight"), self.sd, "lm_head", self.rnd)

    def forward(self, inputs_embeds: Tensor, labels: Optional[Tensor] = None):
        """Teacher-forced pass (models/custom_salmon.py:630-636): returns (logits [B,T,V], loss or None)."""
        B, T, _ = inputs_embeds.shape
        pos = torch.arange(T)[None].expand(B, T)
        lg = self.logits(self.forward_hidden(inputs_embeds.float(), pos))
        loss = None
        if labels is not None:  # HF shift-by-one CE, ignore_index=-100, mean over valid targets
            loss = F.cross_entropy(lg[:, :-1].reshape(-1, lg.shape[-1]), labels[:, 1:].reshape(-1), ignore_index=-100)
        return lg, loss

    def generate_greedy(self, inputs_embeds: Tensor, max_new_tokens: int, eos_id: int, pad_id: int,
                        return_first_logits: bool = False):
        """HF GenerationMixin greedy search with inputs_embeds only (models/custom_salmon.py:704-720):
        returns only the new tokens [B, <=max_new_tokens]; finished rows are filled with pad_id; stops when all
        rows have emitted EOS (min_length is a no-op with inputs_embeds)."""
        B, T, _ = inputs_embeds.shape
        cache: list = []
        pos = torch.arange(T)[None].expand(B, T)
        h = self.forward_hidden(inputs_embeds.float(), pos, cache)
        lg = self.logits(h[:, -1:])[:, 0]
        first = lg.clone()
        finished = torch.zeros(B, dtype=torch.bool)
        out = []
        for step in range(max_new_tokens):
            tok = lg.argmax(-1)
            tok = torch.where(finished, torch.full_like(tok, pad_id), tok)
            out.append(tok)
            finished = finished | torch.tensor([int(t) in _eos_set(eos_id) for t in tok])
            if bool(finished.all()) or step == max_new_tokens - 1:
                break
            e = self.embed(tok)[:, None]
            p = torch.full((B, 1), T + step)
            lg = self.logits(self.forward_hidden(e, p, cache))[:, 0]
        ids = torch.stack(out, dim=1)
        return (ids, first) if return_first_logits else ids

    def generate_beam(self, inputs_embeds: Tensor, max_new_tokens: int, eos_id: int, pad_id: int, num_beams: int,
                      length_penalty: float = 1.0, return_scores: bool = False, repetition_penalty: float = 1.0):
        """HF beam search as ``generate(inputs_embeds=..., num_beams=K, length_penalty=...)`` runs it for
        models/custom_salmon.py:704-715 / models/multi_task_model.py:142 (bookkeeping: ``BeamBookkeeping``).  Returns the best
        finished hypothesis per row [B, width] padded with ``pad_id`` (width = longest returned)."""
        Bn, T, _ = inputs_embeds.shape
        K = num_beams
        cache: list = []
        h = self.forward_hidden(inputs_embeds.float(), torch.arange(T)[None].expand(Bn, T), cache)
        lg = self.logits(h[:, -1:])[:, 0]                                             # [B, V]
        cache = [(k.repeat_interleave(K, 0), v.repeat_interleave(K, 0)) for k, v in cache]
        lg = lg.repeat_interleave(K, 0)                                               # beam-major rows: b * K + k
        bk = BeamBookkeeping(Bn, K, max_new_tokens, eos_id, length_penalty, repetition_penalty)
        for step in range(max_new_tokens):
            parents, toks = bk.step(lg)
            if bk.done:
                break
            cache = [(k[parents], v[parents]) for k, v in cache]
            lg = self.logits(self.forward_hidden(self.embed(toks)[:, None], torch.full((Bn * K, 1), T + step), cache))[:, 0]
        out, scores = bk.result(pad_id)
        return (out, scores) if return_scores else out

    def teacher_forced_logits(self, inputs_embeds: Tensor, tokens: Tensor) -> Tensor:
        """Logits the greedy loop of ``generate_greedy`` would see if its choices were ``tokens`` [B, n]: prefill, then feed
        tokens[:, t] at position T + t.  Returns [B, n, V]: row t is the distribution token t is drawn from.  Used to check
        another implementation's greedy ids decision by decision (each must be this oracle's arg-max up to the logit error),
        without the two paths diverging after the first numerical coin flip."""
        B, T, _ = inputs_embeds.shape
        n = tokens.shape[1]
        cache: list = []
        h = self.forward_hidden(inputs_embeds.float(), torch.arange(T)[None].expand(B, T), cache)
        out = [self.logits(h[:, -1:])[:, 0]]
        for step in range(n - 1):
            e = self.embed(tokens[:, step])[:, None]
            out.append(self.logits(self.forward_hidden(e, torch.full((B, 1), T + step), cache))[:, 0])
        return torch.stack(out, dim=1)

    def generate_sampled(self, inputs_embeds: Tensor, max_new_tokens: int, eos_id: int, pad_id: int, uniforms: Tensor,
                         do_sample: bool = True, temperature: float = 1.0, top_k: int = 50, top_p: float = 1.0,
                         repetition_penalty: float = 1.0):
        """HF sample mode with inputs_embeds only (models/custom_salmon.py:705-721): per step the scores go through
        ``sample_filter`` and the token is the inverse-CDF draw ``sample_pick(probs, uniforms[step, b])``.  With
        ``do_sample=False`` it is greedy search under the repetition penalty (temperature / top-k / top-p ignored, as in HF)."""
        B, T, _ = inputs_embeds.shape
        cache: list = []
        h = self.forward_hidden(inputs_embeds.float(), torch.arange(T)[None].expand(B, T), cache)
        lg = self.logits(h[:, -1:])[:, 0]
        finished = torch.zeros(B, dtype=torch.bool)
        out: list = []
        kept = []
        for step in range(max_new_tokens):
            toks = []
            for b in range(B):
                prev = [int(o[b]) for o in out]
                if do_sample:
                    ids, probs = sample_filter(lg[b].numpy(), prev, repetition_penalty, temperature, top_k, top_p)
                    toks.append(int(ids[sample_pick(probs, float(uniforms[step, b]))]))
                else:
                    ids, probs = sample_filter(lg[b].numpy(), prev, repetition_penalty, 1.0, 1, 1.0)
                    toks.append(int(ids[0]))
                if step == 0:
                    kept.append((ids, probs))
            tok = torch.where(finished, torch.full((B,), pad_id), torch.tensor(toks))
            out.append(tok)
            finished = finished | torch.tensor([int(t) in _eos_set(eos_id) for t in tok])
            if bool(finished.all()) or step == max_new_tokens - 1:
                break
            lg = self.logits(self.forward_hidden(self.embed(tok)[:, None], torch.full((B, 1), T + step), cache))[:, 0]
        return torch.stack(out, dim=1), kept


def _eos_set(eos_id) -> frozenset:
    """HF takes one EOS id or a list of them (generation_config.eos_token_id); negative ids never match."""
    ids = eos_id if isinstance(eos_id, (tuple, list)) else [eos_id]
    return frozenset(int(e) for e in ids if int(e) >= 0)


class BeamBookkeeping:
    """The scorer of HF beam search (transformers/generation/utils.py `_beam_search`, early_stopping=False, one EOS id,
    do_sample=False, prompt length 0 as with inputs_embeds), one ``step(logits [B*K, V])`` at a time:
      * 2K continuations (3K with two EOS ids) are kept per row (best accumulated log-probability first);
      * the K best of them that do not stop (EOS or the length limit) run on — a stopping one stays in line at score - 1e9;
      * those among the first K that stop compete for the K finished slots at sum / len**length_penalty, but only while the
        row is open; the row closes once its best running score / cur_len**length_penalty no longer beats its worst
        finished slot (all K slots filled);
      * the search ends when every row is closed or every continuation stopped (length limit)."""

    def __init__(self, Bn: int, K: int, max_new_tokens: int, eos_id, length_penalty: float, repetition_penalty: float = 1.0):
        self.Bn, self.K, self.T, self.eos, self.lp = Bn, K, max_new_tokens, _eos_set(eos_id), length_penalty
        self.rep = float(repetition_penalty)
        self.keep = max(2, 1 + len(self.eos)) * K           # continuations kept per row: (1 + number of EOS ids) * K, at least 2K
        self.neg = torch.tensor(-1.0e9, dtype=torch.float32)
        self.run_seq = [[[] for _ in range(K)] for _ in range(Bn)]
        self.run_score = torch.full((Bn, K), -1.0e9, dtype=torch.float32)
        self.run_score[:, 0] = 0.0
        self.fin = [[(self.neg.clone(), [], False) for _ in range(K)] for _ in range(Bn)]      # (score, tokens, finished)
        self.open = [True] * Bn
        self.t = 0
        self.done = False

    def step(self, lg: Tensor):
        """-> (parent row index b*K + k, token) per running beam [B*K] for the next forward pass."""
        Bn, K, NEG, step = self.Bn, self.K, self.neg, self.t
        V = lg.shape[-1]
        logp = F.log_softmax(lg.float(), dim=-1).view(Bn, K, V)
        if self.rep != 1.0:      # RepetitionPenaltyLogitsProcessor on the LOG-PROBABILITIES of each beam's own tokens (HF applies the
            logp = logp.clone()  # processors after log_softmax in beam search): x < 0 ? x * penalty : x / penalty
            for b in range(Bn):
                for k in range(K):
                    for t in set(self.run_seq[b][k]):
                        x = logp[b, k, t]
                        logp[b, k, t] = x * self.rep if x < 0 else x / self.rep
        acc = (logp + self.run_score[:, :, None]).view(Bn, K * V)
        top_v, top_i = torch.sort(acc, dim=1, descending=True, stable=True)      # torch.topk, with ties to the lower beam*V + token
        top_v, top_i = top_v[:, : self.keep], top_i[:, : self.keep]
        parents = torch.zeros(Bn, K, dtype=torch.long)
        toks = torch.zeros(Bn, K, dtype=torch.long)
        lenpen = float(step + 1) ** self.lp
        all_stop = True
        for b in range(Bn):
            cand = []
            for j in range(self.keep):
                par, tok = int(top_i[b, j]) // V, int(top_i[b, j]) % V
                stops = (tok in self.eos) or (step + 1 >= self.T)
                cand.append((top_v[b, j], par, tok, stops, self.run_seq[b][par] + [tok]))
            all_stop = all_stop and all(c[3] for c in cand)
            order = sorted(range(self.keep), key=lambda j: (-float(cand[j][0] + (NEG if cand[j][3] else 0.0)), j))[:K]
            merged = list(self.fin[b])
            for j, (v, par, tok, stops, seq) in enumerate(cand):
                sc = v / lenpen
                if not self.open[b]:
                    sc = sc + NEG
                just = stops and j < K
                if not just:
                    sc = sc + NEG
                merged.append((sc, seq, just))
            keep = sorted(range(len(merged)), key=lambda j: (-float(merged[j][0]), j))[:K]
            self.fin[b] = [merged[j] for j in keep]
            self.run_seq[b] = [cand[j][4] for j in order]
            for i, j in enumerate(order):
                self.run_score[b, i] = cand[j][0] + (NEG if cand[j][3] else 0.0)
                parents[b, i], toks[b, i] = cand[j][1], cand[j][2]
            best = self.run_score[b, 0] / lenpen
            worst = min(f[0] for f in self.fin[b])
            self.open[b] = self.open[b] and any(bool(best > (worst if f[2] else NEG)) for f in self.fin[b])
        self.t += 1
        self.done = self.done or not any(self.open) or all_stop
        return (parents + torch.arange(Bn)[:, None] * K).reshape(-1), toks.reshape(-1)

    def result(self, pad_id: int):
        width = max(len(self.fin[b][0][1]) for b in range(self.Bn))
        out = torch.full((self.Bn, width), pad_id, dtype=torch.long)
        for b in range(self.Bn):
            out[b, : len(self.fin[b][0][1])] = torch.tensor(self.fin[b][0][1], dtype=torch.long)
        return out, torch.stack([self.fin[b][0][0] for b in range(self.Bn)])


def sample_filter(logits, prev_tokens, repetition_penalty: float, temperature: float, top_k: int, top_p: float):
    """The distribution HF samples from, restated from the logits processors the reference's ``generate`` call builds
    (transformers/generation/logits_process.py: RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
    TopPLogitsWarper) in float32: returns (token ids, probabilities) of the kept tokens ordered by (score descending,
    token ascending) — the order the HIP kernel draws in."""
    x = np.asarray(logits, dtype=np.float32).copy()
    raw = x.copy()
    if repetition_penalty != 1.0:
        pen = np.float32(repetition_penalty)
        for t in set(int(t) for t in prev_tokens):
            if 0 <= t < x.shape[0]:
                x[t] = raw[t] * pen if raw[t] < 0 else raw[t] / pen
    x = x / np.float32(temperature)
    kth = np.sort(x)[::-1][min(top_k, x.shape[0]) - 1]
    cand = np.nonzero(x >= kth)[0]
    order = np.lexsort((cand, -x[cand].astype(np.float64)))
    cand = cand[order]
    e = np.exp((x[cand] - x[cand[0]]).astype(np.float32)).astype(np.float32)
    p = e / np.float32(e.sum(dtype=np.float32))
    keep = len(cand)
    if top_p < 1.0:
        keep, tail = 1, np.float32(0.0)
        for j in range(len(cand) - 1, 0, -1):
            tail = np.float32(tail + p[j])
            if tail > np.float32(1.0) - np.float32(top_p):
                keep = j + 1
                break
    total = np.float32(0.0)
    for j in range(keep):
        total = np.float32(total + p[j])
    return cand[:keep].astype(np.int64), (p[:keep] / total).astype(np.float32)


def sample_pick(probs, u: float) -> int:
    """First index whose running probability mass exceeds ``u`` (inverse CDF over the kept, renormalised distribution)."""
    acc = 0.0
    for j, pj in enumerate(probs):
        acc += float(pj)
        if acc > u:
            return j
    return len(probs) - 1

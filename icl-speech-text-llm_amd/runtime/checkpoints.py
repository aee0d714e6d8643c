"""Checkpoint ingestion: on-disk formats → the canonical parameter names the packers read.

What the reference loads, and where (all through third-party loaders):
  * ``llama_path``   HF Llama / Vicuna folder → ``LlamaForCausalLM.from_pretrained`` + ``resize_token_embeddings`` for the added
                     ``[PAD]`` token                                                    (models/custom_salmon.py:30,64-97)
  * ``whisper_path`` HF Whisper folder → ``WhisperModel.from_pretrained(...).encoder``  (:31)
  * ``beats_path``   BEATs ``.pt`` with ``{"cfg", "model"}``                           (:32)
  * ``ckpt_path``    ``salmonn_v1.pth`` → ``{"model": {...}}`` holding Q-Former, projector, LN and LoRA tensors  (:47)
  * ``--peft_model_path`` fine-tuned trainable-only checkpoints, 4 key conventions       (inference/inference.py:157-177)
  * Qwen2-Audio HF folder                                                                (models/custom_qwen.py:51-55)
Here the files are read directly (``safetensors`` / ``torch.load``), renamed onto the canonical names
(``llama_model.*``, ``speech_encoder.*``, ``beats.*`` …, see ``runtime/synth.py``) and handed to ``load_state_dict``; the
packers then build the fused bf16 layouts.  Architecture numbers come from the folder's ``config.json``.
"""
from __future__ import annotations

import glob
import json
import os
from dataclasses import replace
from typing import Dict, Optional

import torch

from .config import LlamaCfg, WhisperCfg

SD = Dict[str, torch.Tensor]


def read_tensors(path: str) -> SD:
    """All tensors of a checkpoint file or HF model folder (sharded or not), on the CPU."""
    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            files = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors or pytorch_model*.bin under {path}")
    else:
        files = [path]
    out: SD = {}
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            out.update(load_file(f, device="cpu"))
        else:
            obj = torch.load(f, map_location="cpu", weights_only=False)
            for wrapper in ("model", "state_dict", "model_state_dict"):
                if isinstance(obj, dict) and wrapper in obj and isinstance(obj[wrapper], dict):
                    obj = obj[wrapper]
                    break
            out.update({k: v for k, v in obj.items() if isinstance(v, torch.Tensor)})
    return out


def read_config(path: str) -> Optional[dict]:
    f = os.path.join(path, "config.json") if os.path.isdir(path) else None
    if f and os.path.isfile(f):
        with open(f) as fh:
            return json.load(fh)
    return None


# ---- Llama / Vicuna -----------------------------------------------------------------------------------------------
def llama_cfg_from_hf(cfg_json: dict, base: LlamaCfg) -> LlamaCfg:
    """``config.json`` of a HF Llama folder → LlamaCfg (vocab grows by the ``[PAD]`` token SALMONN adds)."""
    kv = cfg_json.get("num_key_value_heads", cfg_json["num_attention_heads"])
    if kv != cfg_json["num_attention_heads"]:
        raise NotImplementedError("grouped-query attention checkpoints are not supported by the decode kernels yet")
    return replace(base, hidden=cfg_json["hidden_size"], n_layers=cfg_json["num_hidden_layers"],
                   n_heads=cfg_json["num_attention_heads"], ffn=cfg_json["intermediate_size"],
                   vocab=cfg_json["vocab_size"] + 1, rms_eps=cfg_json.get("rms_norm_eps", base.rms_eps),
                   rope_theta=float((cfg_json.get("rope_parameters") or {}).get("rope_theta", cfg_json.get("rope_theta", base.rope_theta))),
                   max_pos=cfg_json.get("max_position_embeddings", base.max_pos),
                   bos_id=cfg_json.get("bos_token_id", base.bos_id), eos_id=cfg_json.get("eos_token_id", base.eos_id),
                   pad_id=cfg_json["vocab_size"])


def llama_from_hf(sd: SD, vocab: int) -> SD:
    """HF ``LlamaForCausalLM`` names → ``llama_model.*``; embedding and lm_head grow to ``vocab`` rows (new rows zero: the
    pad token is only ever written into finished rows, never fed back or scored)."""
    out: SD = {}
    for k, v in sd.items():
        if k.endswith("rotary_emb.inv_freq"):
            continue
        out["llama_model." + k] = v
    if "llama_model.lm_head.weight" not in out:                       # tied embeddings
        out["llama_model.lm_head.weight"] = out["llama_model.model.embed_tokens.weight"]
    for key in ("llama_model.model.embed_tokens.weight", "llama_model.lm_head.weight"):
        w = out[key]
        if w.shape[0] < vocab:
            out[key] = torch.cat([w, torch.zeros(vocab - w.shape[0], w.shape[1], dtype=w.dtype)], 0)
        elif w.shape[0] > vocab:
            raise ValueError(f"{key} has {w.shape[0]} rows, the configuration expects {vocab}")
    return out


# ---- Whisper encoder ----------------------------------------------------------------------------------------------
def whisper_cfg_from_hf(cfg_json: dict) -> WhisperCfg:
    return WhisperCfg(d_model=cfg_json["d_model"], n_layers=cfg_json["encoder_layers"],
                      n_heads=cfg_json["encoder_attention_heads"], ffn=cfg_json["encoder_ffn_dim"],
                      n_mels=cfg_json.get("num_mel_bins", 80), n_ctx=cfg_json.get("max_source_positions", 1500))


def whisper_from_hf(sd: SD) -> SD:
    """HF ``WhisperModel`` / ``WhisperForConditionalGeneration`` names → ``speech_encoder.*`` (decoder tensors dropped)."""
    out: SD = {}
    for k, v in sd.items():
        for prefix in ("model.encoder.", "encoder."):
            if k.startswith(prefix):
                out["speech_encoder." + k[len(prefix):]] = v
                break
    if not out:
        raise KeyError("no Whisper encoder tensors ('model.encoder.*' / 'encoder.*') in the checkpoint")
    return out


# ---- BEATs ---------------------------------------------------------------------------------------------------------
def beats_from_pt(obj) -> SD:
    """BEATs release checkpoint ``{"cfg": …, "model": state_dict}`` → ``beats.*`` (the classifier head is dropped)."""
    sd = obj["model"] if isinstance(obj, dict) and "model" in obj else obj
    return {"beats." + k: v for k, v in sd.items() if not k.startswith("predictor")}


def load_pretrained_parts(llama_path: str = "", whisper_path: str = "", beats_path: str = "", vocab: Optional[int] = None) -> SD:
    """Everything that exists on disk among the three pretrained parts, under canonical names."""
    out: SD = {}
    if llama_path and os.path.isdir(llama_path):
        sd = read_tensors(llama_path)
        cj = read_config(llama_path)
        out.update(llama_from_hf(sd, vocab if vocab is not None else (cj["vocab_size"] + 1 if cj else sd["model.embed_tokens.weight"].shape[0])))
    if whisper_path and os.path.isdir(whisper_path):
        out.update(whisper_from_hf(read_tensors(whisper_path)))
    if beats_path and os.path.isfile(beats_path):
        out.update(beats_from_pt(torch.load(beats_path, map_location="cpu", weights_only=False)))
    return out

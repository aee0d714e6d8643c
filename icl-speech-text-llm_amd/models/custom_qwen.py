"""CustomQwen on MI355X — same plugin surface as the reference's ``models/custom_qwen.py`` (:29-247), with
``Qwen2AudioForConditionalGeneration`` replaced by ``runtime.QwenAudioRuntime`` (HIP kernels behind the C-ABI).

Reference behaviour kept: constructor kwargs (:29-40); ``forward`` builds ``labels = -100`` up to ``prompt_length`` and the
input ids after it (:142-145) and returns {"loss", "logits", "labels"} (:186-197); ``generate_output`` generates 10 new
tokens and decodes only them with ``skip_special_tokens=True`` (:228-246); ``get_speech_embeddings`` returns four Nones
(:117-124); ``input_processor`` exposes ``.tokenizer`` and ``.batch_decode``.  LoRA (r=8 on q_proj,k_proj, :71-80) is kept
un-merged.  The checkpoint folder's ``generation_config.json`` is inherited as HF's ``from_pretrained`` does (sampling knobs,
beams, the EOS id list); without one (synthetic weights) generation is greedy.  Like the reference's ``generate_output`` (:227-233)
this one reads no generation knob from the batch dict.  Differences: rows of different length may
share a batch (padding is stripped through ``attention_mask`` and the rows are packed); no fp16 autocast (bf16 inside).
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from ..runtime import synth
from ..runtime.config import QwenAudioCfg
from ..utils.tokenization import ByteTokenizer, Encoding
from .base_model import BaseModel
from .custom_salmon import PackedTreeModule

logger = logging.getLogger(__name__)


_GENERATION_KEYS = ("do_sample", "temperature", "top_p", "top_k", "repetition_penalty", "num_beams", "length_penalty",
                    "eos_token_id", "pad_token_id")


def _read_generation_config(model_path) -> Dict[str, Any]:
    """``generation_config.json`` of a local checkpoint folder -> the knobs this path implements (others are logged)."""
    import json
    f = os.path.join(model_path, "generation_config.json") if model_path and os.path.isdir(str(model_path)) else None
    if not f or not os.path.isfile(f):
        return {}
    with open(f) as fh:
        raw = json.load(fh)
    cfg = {k: raw[k] for k in _GENERATION_KEYS if raw.get(k) is not None}
    if isinstance(cfg.get("eos_token_id"), list) and len(cfg["eos_token_id"]) > 2:
        logger.warning("generation_config names %d EOS ids; the first two are used", len(cfg["eos_token_id"]))
        cfg["eos_token_id"] = cfg["eos_token_id"][:2]
    other = sorted(k for k in raw if k not in _GENERATION_KEYS and k not in ("bos_token_id", "transformers_version", "_from_model_config"))
    if other:
        logger.info("generation_config keys without effect on this path: %s", other)
    return cfg


class QwenModule(PackedTreeModule):
    """Parameter tree under HF Qwen2-Audio key names + the packed HIP runtime built from it on first use."""

    def __init__(self, cfg: QwenAudioCfg, device, seed: int = 0):
        super().__init__()
        self.cfg = cfg
        self._init_tree(device, synth.qwen_audio_state(cfg, seed=seed, device=torch.device(device), dtype=torch.bfloat16))

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        from ..runtime.qwen import normalize_qwen_keys
        sd = normalize_qwen_keys(dict(state_dict))
        own = self.state_dict()
        sd = {k: v for k, v in sd.items() if k in own or strict}
        out = super().load_state_dict(sd, strict=strict, assign=assign)
        self._runtime = None
        return out

    def _build_runtime(self):
        from ..runtime.qwen import QwenAudioRuntime
        return QwenAudioRuntime(self.cfg, self.state_dict(), device=self._device)


class QwenSpecialTokenizer(ByteTokenizer):
    """Byte-level fallback tokenizer that knows Qwen2-Audio's control strings (used only when the real tokenizer files
    are not on disk).  The special ids sit at the top of the vocabulary."""

    SPECIALS = ("<|im_start|>", "<|im_end|>", "<|audio_bos|>", "<|audio_eos|>", "<|AUDIO|>")

    def __init__(self, vocab_size: int, audio_token_id: int, eos_id: int, pad_id: int):
        super().__init__(vocab_size)
        self.pad_token_id, self.eos_token_id = pad_id, eos_id
        self.eos_token = "<|im_end|>"
        free = [i for i in range(vocab_size - 1, 258, -1) if i not in (audio_token_id, pad_id, eos_id)]
        self.special_ids = {"<|AUDIO|>": audio_token_id, "<|im_end|>": eos_id}
        for name in self.SPECIALS:
            if name not in self.special_ids:
                self.special_ids[name] = free.pop(0)
        self._special_set = set(self.special_ids.values())

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        out: List[int] = []
        i = 0
        while i < len(text):
            for name, tid in self.special_ids.items():
                if text.startswith(name, i):
                    out.append(tid)
                    i += len(name)
                    break
            else:
                j = i + 1
                while j < len(text) and not any(text.startswith(n, j) for n in self.special_ids):
                    j += 1
                out.extend(b + 3 for b in text[i:j].encode("utf-8"))
                i = j
        return out

    def decode(self, ids, skip_special_tokens: bool = False, **kw) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        if skip_special_tokens:
            ids = [t for t in ids if int(t) not in self._special_set]
        return super().decode(ids, skip_special_tokens=skip_special_tokens)


class QwenHostProcessor:
    """What the reference takes from ``AutoProcessor`` (``.tokenizer``, ``.batch_decode``, ``apply_chat_template`` and the
    text+audio ``__call__``), with the audio feature extraction routed to the K1 kernel (128 mel bins)."""

    def __init__(self, owner: "CustomQwen", tokenizer):
        self._owner, self.tokenizer = owner, tokenizer

    def batch_decode(self, ids, skip_special_tokens: bool = True, **kw):
        return self.tokenizer.batch_decode(ids, skip_special_tokens=skip_special_tokens)

    def apply_chat_template(self, conversation, add_generation_prompt: bool = True, tokenize: bool = False) -> str:
        """Qwen2-Audio's chat template [upstream chat_template.json, restated]: system/user turns in <|im_start|>…<|im_end|>,
        every audio item rendered as ``Audio k: <|audio_bos|><|AUDIO|><|audio_eos|>\\n``."""
        out, n_audio = [], 0
        if conversation and conversation[0]["role"] != "system":
            out.append("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n")
        for msg in conversation:
            out.append(f"<|im_start|>{msg['role']}\n")
            if isinstance(msg["content"], str):
                out.append(msg["content"])
            else:
                for item in msg["content"]:
                    if item.get("type") == "audio" or "audio_url" in item or "audio" in item:
                        n_audio += 1
                        out.append(f"Audio {n_audio}: <|audio_bos|><|AUDIO|><|audio_eos|>\n")
                    elif "text" in item:
                        out.append(item["text"])
            out.append("<|im_end|>\n")
        if add_generation_prompt:
            out.append("<|im_start|>assistant\n")
        return "".join(out)

    def __call__(self, text, audios=None, return_tensors="pt", sampling_rate: int = 16000, **kw) -> Encoding:
        """text (str) with one ``<|AUDIO|>`` per audio -> input_ids with each placeholder expanded to the audio's feature
        count, attention_mask, input_features [n,128,3000] f32, feature_attention_mask [n,3000]."""
        import numpy as np
        from ..runtime.engines import QwenAudioTowerHIP
        audios = list(audios or [])
        mel_lens = [min(3000, -(-len(a) // 160)) for a in audios]
        ids = self.tokenizer.encode(text, add_special_tokens=False)
        aid = self._owner.cfg.audio_token_id
        if sum(t == aid for t in ids) != len(audios):
            raise ValueError("number of <|AUDIO|> placeholders does not match the number of audios")
        out, k = [], 0
        for t in ids:
            if t == aid:
                out.extend([aid] * QwenAudioTowerHIP.output_lengths(mel_lens[k])[1])
                k += 1
            else:
                out.append(t)
        enc = {"input_ids": torch.tensor([out], dtype=torch.long), "attention_mask": torch.ones(1, len(out), dtype=torch.long)}
        if audios:
            L = max(len(a) for a in audios)
            wav = torch.zeros(len(audios), min(L, 480000))
            for i, a in enumerate(audios):
                a = torch.as_tensor(np.asarray(a), dtype=torch.float32)[:480000]
                wav[i, :a.numel()] = a
            enc["input_features"] = self._owner.model.runtime.logmel_features(wav, [min(len(a), 480000) for a in audios]).cpu()
            fm = torch.zeros(len(audios), 3000, dtype=torch.long)
            for i, m in enumerate(mel_lens):
                fm[i, :m] = 1
            enc["feature_attention_mask"] = fm
        return Encoding(enc)


class CustomQwen(BaseModel):
    def __init__(self, model_path: str = "Qwen/Qwen2-Audio-7B-Instruct", lora: bool = True, low_resource: bool = True,
                 lora_rank: int = 8, lora_alpha: int = 32, lora_dropout: float = 0.05, prompt_template: str = "",
                 max_txt_len: int = 512, ckpt_path: Optional[str] = None, device=None, use_fp16: bool = True,
                 arch=None, tokenizer=None, seed: int = 0, generation_config: Optional[Dict[str, Any]] = None, **ignored):
        super().__init__(device=device, use_fp16=use_fp16)
        if ignored:
            logger.info("CustomQwen: ignoring unknown kwargs %s", sorted(ignored))
        from dataclasses import replace
        cfg = arch if isinstance(arch, QwenAudioCfg) else (QwenAudioCfg.tiny(lora=lora) if arch == "tiny" else QwenAudioCfg())
        cfg = replace(cfg, llm=replace(cfg.llm, lora_rank=lora_rank if lora else 0, lora_alpha=float(lora_alpha)))
        self.cfg = cfg
        self.model = QwenModule(cfg, self.device, seed=seed)
        if ckpt_path and os.path.isfile(ckpt_path):
            ckpt = torch.load(ckpt_path, map_location="cpu")
            self.model.load_state_dict(ckpt.get("model", ckpt), strict=False)
        if tokenizer is None:
            if model_path and os.path.isdir(model_path):
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(model_path)
            else:
                logger.warning("model_path %r is not a local directory: synthetic weights + byte-level tokenizer", model_path)
                tokenizer = QwenSpecialTokenizer(cfg.llm.vocab, cfg.audio_token_id, cfg.llm.eos_id, cfg.llm.pad_id)
        self.input_processor = QwenHostProcessor(self, tokenizer)
        self.prompt_template, self.max_txt_len, self.lora = prompt_template, max_txt_len, lora
        # the reference calls generate(max_new_tokens=10) and inherits every other knob from the checkpoint's
        # generation_config.json (custom_qwen.py:227-233); an explicit ``generation_config`` kwarg wins
        self.generation_config = dict(_read_generation_config(model_path), **(generation_config or {}))
        self.batch_counter = 0

    @property
    def runtime(self):
        return self.model.runtime

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {(k[len("model."):] if k.startswith("model.") and not k.startswith("model.audio_tower") and not
               k.startswith("model.language_model") and not k.startswith("model.multi_modal") else k): v
              for k, v in state_dict.items()}
        return self.model.load_state_dict(sd, strict=strict, assign=assign)

    def get_speech_embeddings(self, samples):
        return None, None, None, None

    # ---- shared: batch dict -> packed segments ------------------------------------------------------------
    def _rows_and_audio(self, samples):
        input_ids, att = samples["input_ids"], samples.get("attention_mask")
        if input_ids.dim() == 1:
            input_ids = input_ids[None]
            att = att[None] if att is not None else None
        rows = []
        for b in range(input_ids.shape[0]):
            ids = input_ids[b]
            if att is not None:
                ids = ids[att[b].bool()]
            rows.append(ids.tolist())
        feats, fmask = samples.get("input_features"), samples.get("feature_attention_mask")
        speech, out_lens = None, []
        if feats is not None and feats.numel() > 0:
            if feats.dim() == 2:
                feats, fmask = feats[None], (fmask[None] if fmask is not None else None)
            mel_lens = fmask.sum(-1).tolist() if fmask is not None else [3000] * feats.shape[0]
            speech, out_lens = self.runtime.encode_audio(input_features=feats, mel_lens=mel_lens)
        return rows, self.runtime.segments_from_ids(rows, out_lens), speech, input_ids, att

    def forward(self, samples: Dict[str, Any]) -> Dict[str, Any]:
        rows, segs, speech, input_ids, att = self._rows_and_audio(samples)
        B, S = input_ids.shape
        labels = torch.full((B, S), -100, dtype=torch.long)
        for i, pl in enumerate(samples["prompt_length"]):
            labels[i, int(pl):] = input_ids[i, int(pl):].cpu()
        if att is not None:
            labels[att.cpu() == 0] = -100
        rt = self.runtime
        logits_packed, lens = rt.forward_logits(segs, speech)
        V = logits_packed.shape[1]
        logits = torch.zeros(B, S, V, dtype=torch.float32, device=logits_packed.device)
        shifted = torch.full((sum(lens),), -100, dtype=torch.int32)
        off = 0
        for b, n in enumerate(lens):
            pos = att[b].bool().nonzero().flatten() if att is not None else torch.arange(S)
            logits[b, pos.to(logits.device)] = logits_packed[off:off + n]
            lab = labels[b, pos.cpu()]
            shifted[off:off + n - 1] = lab[1:].to(torch.int32)
            off += n
        loss = rt.cross_entropy(logits_packed, shifted).clone()[0]
        self.batch_counter += 1
        return {"loss": loss, "logits": logits, "labels": labels.to(logits.device)}

    def print_trainable_parameters(self):
        """Reference :104-116: one log line with the count of parameters that require gradients (none on this inference path:
        the tree holds frozen parameters) against all parameters."""
        total = sum(p.numel() for p in self.parameters())
        trainable = sum(p.numel() for p in self.parameters() if p.requires_grad)
        logging.info(f"Trainable params: {trainable} ({100 * trainable / max(total, 1):.2f}% of all params)")

    def generate_ids(self, batch: Dict[str, Any], want_first_logits: bool = False):
        """Batch dict -> ``GenerateResult`` (new token ids, first-step logits on request); see CustomSALMONN.generate_ids."""
        rows, segs, speech, _, _ = self._rows_and_audio(batch)
        # The reference calls generate(max_new_tokens=10) and takes EVERY other knob from the model's generation config
        # (custom_qwen.py:227-233): keys a caller puts in the batch dict — MultiTaskModel's max_new_tokens / num_beams / do_sample /
        # temperature among them — never reach HF there, so they are not read here either.  ``self.generation_config`` (the
        # checkpoint folder's generation_config.json, the constructor kwarg, or assigned afterwards) is the one place to set them,
        # ``max_new_tokens`` included; a sampling ``generator`` (not a reference key) may ride in the batch.
        g = dict(self.generation_config)
        if "generator" in batch:
            g["generator"] = batch["generator"]
        eos = g.get("eos_token_id", self.cfg.llm.eos_id)          # an id or HF's list form (Qwen2-Audio: [151645, 151643])
        res = self.runtime.generate(segs, speech, max_new_tokens=int(g.get("max_new_tokens", 10)),
                                    eos_id=tuple(eos) if isinstance(eos, (list, tuple)) else int(eos),
                                    pad_id=int(g.get("pad_token_id", self.cfg.llm.pad_id)),
                                    do_sample=bool(g.get("do_sample", False)), temperature=float(g.get("temperature", 1.0)),
                                    top_p=float(g.get("top_p", 1.0)), top_k=int(g.get("top_k", 50)),
                                    repetition_penalty=float(g.get("repetition_penalty", 1.0)), generator=g.get("generator"),
                                    want_first_logits=want_first_logits, overlong="drop", num_beams=int(g.get("num_beams", 1)),
                                    length_penalty=float(g.get("length_penalty", 1.0)))
        self.last_dropped_rows = tuple(res.dropped)      # rows over max_pos cost their own utterance only (see CustomSALMONN)
        if res.dropped:
            logger.error("rows %s of this batch were not generated: prompt + new tokens exceed max_pos %d", list(res.dropped),
                         self.cfg.llm.max_pos)
        self.batch_counter += 1
        return res

    def decode_ids(self, tokens) -> List[str]:
        return self.input_processor.batch_decode(tokens, skip_special_tokens=True, clean_up_tokenization_spaces=False)

    def generate_output(self, batch: Dict[str, Any]) -> List[str]:
        return self.decode_ids(self.generate_ids(batch).tokens)

    @classmethod
    def from_config(cls, config):
        return cls(**config)

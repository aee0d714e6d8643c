"""CustomSALMONN on MI355X — same plugin surface as the reference's ``models/custom_salmon.py``, with the
external ``SALMONN`` object replaced by ``runtime.SalmonnRuntime`` (hand-written HIP kernels behind a C-ABI).

Reference behaviour kept (file:line of /root/reference/models/custom_salmon.py):
  * constructor kwargs and defaults (:29-51); ``from_config`` = ``cls(**config)`` (:741-749);
  * ``get_speech_embeddings`` 4-tuple (:301-544); ``encode_speech`` (:546-554);
  * ``custom_prompt_wrap`` part splitting on ``<Example{i}>`` / ``<SpeechHere>`` and the interleave order (:115-299);
  * ``forward`` -> {"loss", "logits", "labels"} with the -100 label mask (:556-640);
  * ``generate_output`` -> greedy ids -> ``batch_decode(skip_special_tokens=True)`` (:642-739);
  * attributes read by callers: ``salmonn``, ``llama_model``, ``llama_tokenizer``, ``speech_placeholder``,
    ``batch_counter`` and — missing in the reference class although inference/inference.py:200 reads it —
    ``input_processor``.
Deliberate differences: the serial B x E ``encode_speech`` loop (:503-519) is one batched kernel chain; prompts of
different lengths in one batch are allowed (the reference's ``torch.stack`` :299 forbids them); the first-batch debug
logging with its ``.item()`` syncs is dropped; SQA batches must carry raw waveforms (the log-mel is computed on the GPU).
"""
from __future__ import annotations

import logging
import os
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from ..runtime import synth
from ..runtime.config import LlamaCfg, QFormerCfg, SalmonnCfg, WhisperCfg, with_llama
from ..runtime.packing import normalize_keys
from ..utils.tokenization import load_llama_tokenizer
from .base_model import BaseModel

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------------------
# host-side prompt logic (pure Python: testable without a GPU, pinned to the reference by golden captures)
# ------------------------------------------------------------------------------------------------------
def split_prompt(prompt: str, max_examples: int, has_example_embeds: bool, speech_placeholder: str = "<SpeechHere>",
                 is_sqa: bool = False) -> List[str]:
    """Text parts of one prompt, exactly as custom_prompt_wrap cuts them (reference :131-175).  ``is_sqa`` is the
    reference's ``isinstance(embeds, tuple)``: exemplars are cut at ``<Document{i}>`` then ``<Question{i}>`` (two parts
    each); the query is cut at ``<Document>`` then ``<Question>`` whenever ``<Question>`` occurs in what is left."""
    parts: List[str] = []
    suffix = prompt
    if max_examples > 0 and has_example_embeds:
        for i in range(max_examples):
            if is_sqa:
                q_marker, d_marker = f"<Question{i}>", f"<Document{i}>"
                if q_marker in suffix and d_marker in suffix:
                    before_d, rest = suffix.split(d_marker, 1)
                    middle, suffix = rest.split(q_marker, 1)
                    parts += [before_d, middle]
                continue
            marker = f"<Example{i}>"
            if marker in suffix:
                before, suffix = suffix.split(marker, 1)
                parts.append(before)
            else:
                parts.append("")
    if "<Question>" in suffix:
        before_d, rest = suffix.split("<Document>", 1)
        middle, suffix = rest.split("<Question>", 1)
        parts += [before_d, middle]
    elif speech_placeholder in suffix:
        before, suffix = suffix.split(speech_placeholder)   # ValueError on a repeated placeholder, as in the reference
        parts.append(before)
    else:
        parts.append(suffix)
        suffix = ""
    parts.append(suffix)
    return parts


def interleave_plan(n_parts: int, max_examples: int, n_example_embeds: Optional[int], has_main_speech: bool) -> List[Tuple[str, int]]:
    """Order of pieces in the wrapped sequence (reference :243-271): ('text', part index) | ('example', i) | ('speech', 0)."""
    plan: List[Tuple[str, int]] = []
    for i in range(n_parts - 2):
        plan.append(("text", i))
        if i < max_examples and n_example_embeds is not None and i < n_example_embeds:
            plan.append(("example", i))
    if has_main_speech:
        plan += [("text", n_parts - 2), ("speech", 0), ("text", n_parts - 1)]
    elif n_parts >= 2:
        plan += [("text", n_parts - 2), ("text", n_parts - 1)]
    else:
        plan.append(("text", n_parts - 1))
    return plan


def interleave_plan_sqa(n_parts: int, max_examples: int, n_example_embeds: Optional[int], has_main_speech: bool) -> List[Tuple[str, int]]:
    """SQA order (reference :206-241): per exemplar ``text 2i, document audio, text 2i+1, question audio`` (the text between
    them is dropped with the audios when the row has no embedding for exemplar i), then
    ``text[-3], document, text[-2], question, text[-1]``.  Kinds: 'text' | 'example_d' | 'example_q' | 'speech_d' | 'speech_q'."""
    plan: List[Tuple[str, int]] = []
    for i in range(max_examples):
        plan.append(("text", 2 * i))
        if n_example_embeds is not None and i < n_example_embeds:
            plan += [("example_d", i), ("text", 2 * i + 1), ("example_q", i)]
    if has_main_speech:
        plan += [("text", n_parts - 3), ("speech_d", 0), ("text", n_parts - 2), ("speech_q", 0), ("text", n_parts - 1)]
    else:
        plan += [("text", n_parts - 3), ("text", n_parts - 2), ("text", n_parts - 1)]
    return plan


def build_labels(prompt_len: int, target_ids: torch.Tensor, target_mask: torch.Tensor) -> torch.Tensor:
    """labels = -100 over the prompt, target ids over the completion, -100 where the completion is padding (reference :617-627)."""
    B, T = target_ids.shape
    labels = torch.full((B, prompt_len + T), -100, dtype=torch.long)
    labels[:, prompt_len:] = target_ids
    labels[:, prompt_len:][target_mask == 0] = -100
    return labels


class _Node(nn.Module):
    """Anonymous container so that parameters keep their checkpoint key names (``a.b.0.c.weight``)."""


def _build_tree(root: nn.Module, tensors: Dict[str, torch.Tensor]):
    for key, t in tensors.items():
        *path, leaf = key.split(".")
        node = root
        for p in path:
            if not hasattr(node, p):
                node.add_module(p, _Node())
            node = getattr(node, p)
        node.register_parameter(leaf, nn.Parameter(t, requires_grad=False))


class PackedTreeModule(nn.Module):
    """A parameter tree under the checkpoint's own key names + the packed HIP runtime built from it on first use.

    Residency: the runtime holds the weights in the layouts the kernels want (fused QKV / gate-up, decode-packed copies);
    once it is built the tree is the *checkpoint image* only (``state_dict`` / ``load_state_dict`` / ``save_checkpoint``), so it
    is moved to host memory and the GPU keeps ONE set of weights per layout in use (the tree + packed + decode-
    packed used to sit on the device together, 40 GB at 7B).  Tensors the runtime shares with the tree without a copy (o_proj,
    down_proj, embedding, lm_head are used in their checkpoint layout) stay alive on the device through the runtime's own
    references.  ``.to(device)`` after that only re-targets the runtime; the tree stays on the host."""

    offload_tree = True

    @staticmethod
    def _indexed(device) -> torch.device:
        """``cuda`` -> ``cuda:<current>``: device comparisons below must not see "cuda" and "cuda:0" as different places."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return device

    def _init_tree(self, device, tensors):
        self._device = self._indexed(device)
        _build_tree(self, tensors)
        self._runtime = None

    def _build_runtime(self):
        raise NotImplementedError

    def _apply(self, fn, *a, **kw):
        probe = fn(torch.empty(0, dtype=torch.bfloat16, device=self._device))
        target = self._indexed(probe.device)
        if probe.dtype != torch.bfloat16:
            # .half() / .float() / .to(dtype): the packed runtime computes in bf16 with f32 accumulation whatever the tree holds
            # (the north-star precision); say so instead of silently ignoring the request
            logger.warning("%s: dtype change to %s ignored — the HIP path keeps bf16 weights and f32 accumulation",
                           type(self).__name__, probe.dtype)
            if target == self._device:
                return self
            fn = lambda t: t.to(target)          # noqa: E731  (keep the move, drop the cast)
        if self._runtime is not None and target == self._device:
            return self                          # nothing to move: the runtime already lives there
        if target.type == "cuda" and any(p.device.type == "cpu" for p in self.parameters()):
            self._device, self._runtime = target, None      # host-resident tree: packed straight to the new device
            return self
        out = super()._apply(fn, *a, **kw)
        self._runtime = None
        self._device = target
        return out

    @property
    def runtime(self):
        if self._runtime is None:
            self._runtime = self._build_runtime()
            if self.offload_tree and self._device.type == "cuda":
                for prm in self.parameters():
                    if prm.device.type == "cuda":
                        prm.data = prm.data.to("cpu")
        return self._runtime


class SalmonnModule(PackedTreeModule):
    """Stand-in for ``SALMONN.models.salmonn_org.SALMONN``: a parameter tree under the checkpoint's own key names
    (``speech_encoder.*``, ``beats.*``, ``ln_speech``, ``ln_audio``, ``speech_Qformer.bert.*``, ``speech_query_tokens``,
    ``speech_llama_proj.*``, ``llama_model.*``) plus the packed HIP runtime built from it on first use."""

    def __init__(self, cfg: SalmonnCfg, device, seed: int = 0):
        super().__init__()
        self.cfg = cfg
        self._init_tree(device, synth.salmonn_state(cfg, seed=seed, device=torch.device(device), dtype=torch.bfloat16))

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = normalize_keys(dict(state_dict))
        own = self.state_dict()
        sd = {k: v for k, v in sd.items() if k in own or strict}
        result = super().load_state_dict(sd, strict=strict, assign=assign)
        self._runtime = None   # re-pack on next use
        return result

    def _build_runtime(self):
        from ..runtime.salmonn import SalmonnRuntime
        return SalmonnRuntime(self.cfg, self.state_dict(), device=self._device)

    def encode_speech(self, spectrogram=None, raw_wav=None, audio_padding_mask=None):
        """SALMONN.encode_speech signature (reference call: custom_salmon.py:550-554) -> (embeds [B,88,H], atts [B,88])."""
        lens = None
        if raw_wav is not None and audio_padding_mask is not None:
            lens = (~audio_padding_mask.bool()).sum(dim=1).tolist()
        padded = [raw_wav.shape[1]] * raw_wav.shape[0] if raw_wav is not None else None
        emb = self.runtime.encode_speech(raw_wav, lens, spectrogram=spectrogram, padded_lens=padded)
        return emb, torch.ones(emb.shape[:2], dtype=torch.long, device=emb.device)


class HipLogMelFeatureExtractor:
    """``WhisperFeatureExtractor``-compatible callable backed by the K1 HIP kernel (main process only: it touches the GPU)."""

    def __init__(self, owner: "CustomSALMONN"):
        self._owner = owner
        self.sampling_rate, self.feature_size, self.n_fft, self.hop_length = 16000, owner.cfg.whisper.n_mels, 400, 160

    def __call__(self, audio, sampling_rate: int = 16000, return_tensors: str = "pt", **_):
        import numpy as np
        if sampling_rate != 16000:
            raise ValueError("the Whisper front-end expects 16 kHz audio")
        wav = torch.as_tensor(np.asarray(audio), dtype=torch.float32)
        if wav.dim() == 1:
            wav = wav[None]
        n = min(wav.shape[1], 480000)
        spec = self._owner.salmonn.runtime.log_mel(wav[:, :480000], [n] * wav.shape[0]).cpu()
        from ..utils.tokenization import Encoding
        return Encoding({"input_features": spec if return_tensors == "pt" else spec.numpy()})


def _arch_from_paths(llama_path: str, whisper_path: str, beats_path: str, arch) -> SalmonnCfg:
    if isinstance(arch, SalmonnCfg):
        return arch
    from ..runtime import checkpoints as ck
    lj = ck.read_config(llama_path) if llama_path else None
    wj = ck.read_config(whisper_path) if whisper_path else None
    if arch is None and (lj or wj):      # architecture numbers from the HF folders' config.json
        base = SalmonnCfg.llama2_7b()
        whisper = ck.whisper_cfg_from_hf(wj) if wj else base.whisper
        llama = ck.llama_cfg_from_hf(lj, base.llama) if lj else base.llama
        beats = base.beats if beats_path else None
        qf = QFormerCfg(enc_width=whisper.d_model + (beats.d_model if beats else 0))
        return SalmonnCfg(whisper=whisper, beats=beats, qformer=qf, llama=llama)
    name = (arch or "").lower()
    if name == "tiny":
        return SalmonnCfg.tiny(use_beats=bool(beats_path))
    base = SalmonnCfg.llama2_13b() if ("13b" in name or (not name and "13b" in llama_path.lower())) else SalmonnCfg.llama2_7b()
    if "tiny" in whisper_path.lower():
        base = SalmonnCfg(whisper=WhisperCfg(d_model=384, n_layers=4, n_heads=6, ffn=1536), beats=base.beats,
                          qformer=base.qformer, llama=base.llama)
    if not beats_path:
        base = SalmonnCfg(whisper=base.whisper, beats=None,
                          qformer=QFormerCfg(enc_width=base.whisper.d_model), llama=base.llama)
    elif base.qformer.enc_width != base.whisper.d_model + base.beats.d_model:
        base = SalmonnCfg(whisper=base.whisper, beats=base.beats,
                          qformer=QFormerCfg(enc_width=base.whisper.d_model + base.beats.d_model), llama=base.llama)
    return base


class CustomSALMONN(BaseModel):
    def __init__(self, llama_path: str = "lmsys/vicuna-13b-v1.1", whisper_path: str = "openai/whisper-large-v2",
                 beats_path: str = "/data2/neeraja/neeraja/BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt",
                 use_speech_Qformer: bool = True, freeze_whisper: bool = True, freeze_beats: bool = True,
                 freeze_speech_QFormer: bool = False, num_speech_query_token: int = 1, window_level_Qformer: bool = True,
                 second_per_window: float = 0.333333, second_stride: float = 0.333333, speech_llama_proj_model: str = "",
                 freeze_speech_llama_proj: bool = False, lora: bool = True, lora_rank: int = 8, lora_alpha: int = 16,
                 lora_dropout: float = 0.05, ckpt_path: str = "/data2/neeraja/neeraja/salmonn_v1.pth", device=None,
                 low_resource: bool = False, use_fp16: bool = False, max_txt_len: int = 128,
                 arch=None, tokenizer=None, seed: int = 0, **ignored):
        super().__init__(device=device, use_fp16=use_fp16)
        if ignored:
            logger.info("CustomSALMONN: ignoring unknown kwargs %s", sorted(ignored))  # e.g. use_cache (model_factory.py:142)
        if not (use_speech_Qformer and window_level_Qformer and num_speech_query_token == 1):
            raise NotImplementedError("the MI355X path implements the window-level speech Q-Former with one query token")
        cfg = _arch_from_paths(llama_path, whisper_path, beats_path, arch)
        cfg = with_llama(cfg, lora_rank=lora_rank if lora else 0, lora_alpha=float(lora_alpha))
        if (second_per_window, second_stride) != (cfg.qformer.second_per_window, cfg.qformer.second_stride):
            from dataclasses import replace
            cfg = replace(cfg, qformer=replace(cfg.qformer, second_per_window=second_per_window, second_stride=second_stride))
        self.cfg = cfg
        self.lora = lora
        self.max_txt_len = max_txt_len
        # low_resource (8-bit bitsandbytes in the reference) has no effect: the north star computes in bf16.
        self.salmonn = SalmonnModule(cfg, self.device, seed=seed)
        from ..runtime.checkpoints import load_pretrained_parts
        pretrained = load_pretrained_parts(llama_path, whisper_path, beats_path, vocab=cfg.llama.vocab)
        if pretrained:
            missing = self.salmonn.load_state_dict(pretrained, strict=False)
            logger.info("loaded %d pretrained tensors (llama/whisper/beats folders)", len(pretrained))
        if ckpt_path and os.path.isfile(ckpt_path):
            ckpt = torch.load(ckpt_path, map_location="cpu")
            self.salmonn.load_state_dict(ckpt.get("model", ckpt), strict=False)
        else:
            logger.warning("checkpoint %r not found: running on seeded synthetic weights", ckpt_path)
        self.speech_tag_start, self.speech_tag_end, self.speech_placeholder = "<Speech>", "</Speech>", "<SpeechHere>"
        # exposed as the reference does (models/custom_salmon.py:108), but NOT registered as a second child module: a child would be
        # walked by .to() on its own and drag the host-resident checkpoint image of the decoder back onto the device
        object.__setattr__(self, "llama_model", self.salmonn.llama_model)
        self.llama_tokenizer = tokenizer if tokenizer is not None else load_llama_tokenizer(llama_path, cfg.llama.vocab)
        self.input_processor = HipLogMelFeatureExtractor(self)
        # knobs the reference's generate call leaves to HF (the Llama folder's generation_config.json, else HF's defaults): top_k only
        self.hf_generation_defaults = {"top_k": 50}
        gc_file = os.path.join(str(llama_path), "generation_config.json") if llama_path and os.path.isdir(str(llama_path)) else None
        if gc_file and os.path.isfile(gc_file):
            import json
            with open(gc_file) as fh:
                self.hf_generation_defaults.update({k: v for k, v in json.load(fh).items() if k == "top_k" and v})
        self.batch_counter = 0

    # ---- nn.Module plumbing -----------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = {(k[len("salmonn."):] if k.startswith("salmonn.") else k): v for k, v in state_dict.items()}
        return self.salmonn.load_state_dict(sd, strict=strict, assign=assign)

    def to(self, *args, **kwargs):
        out = super().to(*args, **kwargs)
        self.device = self.salmonn._device       # where the runtime lives (the parameter tree may sit on the host)
        return out

    @property
    def runtime(self):
        return self.salmonn.runtime

    # ---- speech -----------------------------------------------------------------------------------------
    def encode_speech(self, spectrogram, raw_wav=None, audio_padding_mask=None):
        return self.salmonn.encode_speech(spectrogram=spectrogram, raw_wav=raw_wav, audio_padding_mask=audio_padding_mask)

    @staticmethod
    def _lengths(wavs: torch.Tensor, samples: Dict[str, Any], len_key: str, mask_key: str) -> List[int]:
        """Valid samples per waveform: the batch's length vector when it carries one (this build's collate always does), else
        the reference's boolean padding mask (:407-441 read only the mask), else the full width.  The mask is looked up LAST:
        a ``CollatedBatch`` derives it on demand, and asking for it would build a [B, L] tensor only to count it again."""
        lens = samples.get(len_key)
        if lens is not None:
            flat = lens.reshape(-1).tolist() if isinstance(lens, torch.Tensor) else list(lens)
            return [int(x) for x in flat]
        mask = samples.get(mask_key)
        if mask is not None:
            return (~mask.bool()).sum(dim=-1).reshape(-1).tolist()
        return [wavs.shape[-1]] * int(wavs.numel() // wavs.shape[-1])

    def get_speech_embeddings(self, samples: Dict[str, Any]):
        t0 = time.time()
        if any(k in samples for k in ("question_raw_wav", "question_spectrogram", "example_question_wavs")):
            return self._sqa_speech_embeddings(samples)
        main_wav, main_spec = samples.get("raw_wav"), samples.get("spectrogram")
        has_main = main_wav is not None or main_spec is not None
        ex_wav, ex_spec = samples.get("example_wavs"), samples.get("example_spectrograms")
        has_ex = ex_wav is not None or ex_spec is not None
        if not has_main and not has_ex:
            return None, None, None, None
        rt = self.runtime
        H, ntok = self.cfg.llama.hidden, rt.tokens_per_audio
        use_wav = (not has_main or main_wav is not None) and (not has_ex or ex_wav is not None)
        B = (main_wav if main_wav is not None else main_spec).shape[0] if has_main else ex_wav.shape[0]
        owners: List[Tuple[int, int]] = []      # (batch row, -1 = main | example index)
        if use_wav:
            rows, valid, padded = [], [], []
            if has_main:
                lens = self._lengths(main_wav, samples, "wav_lengths", "padding_mask")
                for b in range(B):
                    rows.append(main_wav[b]); valid.append(lens[b]); padded.append(main_wav.shape[1]); owners.append((b, -1))
            if has_ex:
                E = ex_wav.shape[1]
                elens = self._lengths(ex_wav, samples, "example_wav_lengths", "example_padding_masks")
                nex = samples.get("num_examples")
                for b in range(ex_wav.shape[0]):
                    k = int(nex[b]) if nex is not None else E
                    for e in range(min(E, k)):
                        rows.append(ex_wav[b, e]); valid.append(elens[b * E + e]); padded.append(ex_wav.shape[2]); owners.append((b, e))
            if (not has_ex and main_wav.dtype == torch.float32 and main_wav.is_cuda and main_wav.is_contiguous()
                    and main_wav.device == PackedTreeModule._indexed(self.device)):
                wav = main_wav          # the collated device batch IS the kernel's operand: no per-row gather copies
            else:
                L = max(r.shape[0] for r in rows)
                wav = torch.zeros(len(rows), L, dtype=torch.float32, device=self.device)
                for i, r in enumerate(rows):
                    wav[i, :r.shape[0]] = r.to(device=self.device, dtype=torch.float32)
            emb = rt.encode_speech(wav, valid, padded_lens=padded)
        else:   # caller supplied spectrograms only (no BEATs stream possible)
            specs = []
            if has_main:
                specs += [main_spec[b] for b in range(B)]; owners += [(b, -1) for b in range(B)]
            if has_ex:
                for b in range(ex_spec.shape[0]):
                    for e in range(ex_spec.shape[1]):
                        specs.append(ex_spec[b, e]); owners.append((b, e))
            emb = rt.encode_speech(None, None, spectrogram=torch.stack(specs).to(self.device))
        emb = emb.clone()   # detach from the runtime's reusable workspace
        wins = rt.last_audio_windows      # tokens per audio: ntok (88) unless a clip / its collated width exceeds 30 s
        speech_embeds = speech_atts = example_embeds = example_atts = None
        if has_main:
            wm = wins[0] if use_wav else ntok           # main rows share one collated width, hence one token count
            speech_embeds = emb[:B, :wm]
            speech_atts = torch.ones(B, wm, dtype=torch.long, device=emb.device)
        if has_ex:
            example_embeds = [[] for _ in range(B)]
            example_atts = [[] for _ in range(B)]
            for i, (b, e) in enumerate(owners):
                if e >= 0:
                    example_embeds[b].append(emb[i, :wins[i]])
                    example_atts[b].append(torch.ones(wins[i], dtype=torch.long, device=emb.device))
        logger.debug("Speech embedding generation took %.3f s", time.time() - t0)
        return speech_embeds, speech_atts, example_embeds, example_atts

    def _sqa_speech_embeddings(self, samples: Dict[str, Any]):
        """SQA batches (reference :312-323, :383-404, :444-488): question and document audio of the query and of every speech
        exemplar, all encoded in ONE batched kernel chain; returns ``((q, d), (q_atts, d_atts), [[(q_e, d_e)…]…], atts)``."""
        rt = self.runtime
        q_wav, d_wav = samples.get("question_raw_wav"), samples.get("document_raw_wav")
        eq_wav, ed_wav = samples.get("example_question_wavs"), samples.get("example_document_wavs")
        has_main = q_wav is not None and d_wav is not None
        has_ex = eq_wav is not None and ed_wav is not None
        if not has_main and not has_ex:
            if samples.get("question_spectrogram") is not None:
                raise NotImplementedError("SQA batches must carry raw waveforms: the MI355X path computes the log-mel itself")
            return None, None, None, None
        rows, valid, padded, owners = [], [], [], []
        if has_main:
            B = q_wav.shape[0]
            for side, wav in (("q", q_wav), ("d", d_wav)):
                name = "question" if side == "q" else "document"
                lens = self._lengths(wav, samples, f"{name}_wav_lengths", f"{name}_padding_mask")
                for b in range(B):
                    rows.append(wav[b]); valid.append(lens[b]); padded.append(wav.shape[1]); owners.append((b, side, -1))
        if has_ex:
            B, E = eq_wav.shape[:2]
            nex = samples.get("num_examples")
            for side, wav in (("q", eq_wav), ("d", ed_wav)):
                name = "question" if side == "q" else "document"
                lens = self._lengths(wav, samples, f"example_{name}_wav_lengths", f"example_{name}_padding_masks")
                for b in range(B):
                    for e in range(min(E, int(nex[b]) if nex is not None else E)):
                        rows.append(wav[b, e]); valid.append(lens[b * E + e]); padded.append(wav.shape[2]); owners.append((b, side, e))
        L = max(r.shape[0] for r in rows)
        wav = torch.zeros(len(rows), L, dtype=torch.float32, device=self.device)
        for i, r in enumerate(rows):
            wav[i, :r.shape[0]] = r.to(device=self.device, dtype=torch.float32)
        emb = rt.encode_speech(wav, valid, padded_lens=padded).clone()
        wins = rt.last_audio_windows      # tokens per audio: ntok (88) unless a collated width exceeds 30 s
        speech_embeds = speech_atts = example_embeds = example_atts = None
        if has_main:
            B = q_wav.shape[0]
            wq, wd = wins[0], wins[B]                  # one collated width per side, hence one token count per side
            speech_embeds = (emb[:B, :wq], emb[B:2 * B, :wd])
            speech_atts = (torch.ones(B, wq, dtype=torch.long, device=emb.device), torch.ones(B, wd, dtype=torch.long, device=emb.device))
        if has_ex:
            B = eq_wav.shape[0]
            slot: Dict[Tuple[int, int], Dict[str, torch.Tensor]] = {}
            for i, (b, side, e) in enumerate(owners):
                if e >= 0:
                    slot.setdefault((b, e), {})[side] = emb[i, :wins[i]]
            example_embeds = [[(slot[(b, e)]["q"], slot[(b, e)]["d"]) for e in sorted(k[1] for k in slot if k[0] == b)]
                              for b in range(B)]
            example_atts = [[tuple(torch.ones(t.shape[0], dtype=torch.long, device=emb.device) for t in pair) for pair in row]
                            for row in example_embeds]
        return speech_embeds, speech_atts, example_embeds, example_atts

    # ---- prompt wrap ------------------------------------------------------------------------------------
    def _segments(self, embeds, prompts: Sequence[str], num_examples, example_embeds):
        """Per-row segment lists for the K9 gather + the flat speech-row matrix they index."""
        from ..runtime.salmonn import speech_segment
        max_examples = int(num_examples.max().item()) if num_examples is not None and len(num_examples) else 0
        speech_rows: List[torch.Tensor] = []
        n_rows = 0
        all_segments = []
        is_sqa = isinstance(embeds, tuple)
        for b, prompt in enumerate(prompts):
            parts = split_prompt(prompt, max_examples, example_embeds is not None, self.speech_placeholder, is_sqa=is_sqa)
            ids = [self.llama_tokenizer(p, padding="longest", return_tensors="pt", add_special_tokens=False)["input_ids"]
                   .reshape(-1).tolist() for p in parts]
            n_ex = len(example_embeds[b]) if (example_embeds is not None and b < len(example_embeds)) else None
            segs = []
            plan = (interleave_plan_sqa if is_sqa else interleave_plan)(len(parts), max_examples, n_ex, embeds is not None)
            for kind, i in plan:
                if kind == "text":
                    if ids[i]:
                        segs.append(ids[i])
                else:
                    t = {"speech": lambda: embeds[b], "example": lambda: example_embeds[b][i],
                         "speech_q": lambda: embeds[0][b], "speech_d": lambda: embeds[1][b],
                         "example_q": lambda: example_embeds[b][i][0], "example_d": lambda: example_embeds[b][i][1]}[kind]()
                    speech_rows.append(t)
                    segs.append(speech_segment(n_rows, t.shape[0]))
                    n_rows += t.shape[0]
            all_segments.append(segs)
        speech = torch.cat(speech_rows, dim=0).to(device=self.device, dtype=torch.float32) if speech_rows else None
        return all_segments, speech

    def custom_prompt_wrap(self, embeds, atts, prompts, num_examples=None, example_embeds=None, example_atts=None):
        segs, speech = self._segments(embeds, prompts, num_examples, example_embeds)
        h, lens = self.runtime.embed_prompts(segs, speech, name="wrap_h")
        if len(set(lens)) != 1:
            raise RuntimeError(f"stack expects each tensor to be equal size, but prompts embed to lengths {lens}")
        B, S = len(lens), lens[0]
        return h.view(B, S, -1).clone(), torch.ones(B, S, dtype=torch.long, device=h.device)

    # ---- forward / generate -----------------------------------------------------------------------------
    def forward(self, samples: Dict[str, Any]) -> Dict[str, Any]:
        t0 = time.perf_counter()
        samples = self._host_counts(samples)
        speech_embeds, _, example_embeds, _ = self.get_speech_embeddings(samples)
        t1 = time.perf_counter()
        num_examples = samples.get("num_examples", torch.zeros(len(samples["prompt"]), dtype=torch.long))
        segs, speech = self._segments(speech_embeds, samples["prompt"], num_examples, example_embeds)
        t2 = time.perf_counter()
        tgt = self.llama_tokenizer(samples["completion"], padding="longest", return_tensors="pt",
                                   add_special_tokens=False, return_attention_mask=True)
        tgt_ids, tgt_mask = tgt["input_ids"], tgt["attention_mask"]
        for b, s in enumerate(segs):
            s.append(tgt_ids[b].tolist())
        rt = self.runtime
        logits, lens = rt.forward_logits(segs, speech)
        if len(set(lens)) != 1:
            raise RuntimeError(f"forward() stacks logits over the batch: all rows must embed to one length, got {lens}")
        B, S = len(lens), lens[0]
        T = tgt_ids.shape[1]
        labels = build_labels(S - T, tgt_ids, tgt_mask)
        shifted = torch.full((B, S), -100, dtype=torch.int32)
        shifted[:, :-1] = labels[:, 1:].to(torch.int32)
        loss = rt.cross_entropy(logits, shifted.reshape(-1)).clone()[0]
        self.batch_counter += 1
        return {"loss": loss, "logits": logits.view(B, S, -1).clone(), "labels": labels.to(logits.device)}

    @staticmethod
    def _host_counts(samples: Dict[str, Any]) -> Dict[str, Any]:
        """``num_examples`` on the host BEFORE any kernel of the batch is launched: the prompt logic reads it element by element
        (`_segments`, `get_speech_embeddings`), and reading a device tensor there waits for the speech encoders — the
        tokenisation would then run after them instead of under them."""
        ne = samples.get("num_examples")
        if isinstance(ne, torch.Tensor) and ne.is_cuda:
            samples = dict(samples, num_examples=ne.cpu())
        return samples

    def generate_ids(self, samples: Dict[str, Any], want_first_logits: bool = False):
        """The arithmetic half of ``generate_output``: batch dict -> ``GenerateResult`` (new token ids int64 [B, width] with HF's
        EOS / pad / width rules, and the first-step logits f32 [B, V] on request).  The data-parallel CLI gathers these as
        fixed-shape tensors and decodes on rank 0."""
        t0 = time.perf_counter()
        samples = self._host_counts(samples)
        speech_embeds, _, example_embeds, _ = self.get_speech_embeddings(samples)
        t1 = time.perf_counter()
        num_examples = samples.get("num_examples", torch.zeros(len(samples["prompt"]), dtype=torch.long))
        segs, speech = self._segments(speech_embeds, samples["prompt"], num_examples, example_embeds)
        t2 = time.perf_counter()
        # generation knobs and their defaults as the reference reads them (:708-715); min_length is a no-op with inputs_embeds
        # and length_penalty only acts on beams (num_beams > 1: HF beam search, early_stopping at its default False)
        res = self.runtime.generate(segs, speech, max_new_tokens=int(samples.get("max_new_tokens", 10)),
                                    eos_id=self.llama_tokenizer.eos_token_id, pad_id=self.llama_tokenizer.pad_token_id,
                                    do_sample=bool(samples.get("do_sample", False)),
                                    temperature=float(samples.get("temperature", 0.8)), top_p=float(samples.get("top_p", 0.9)),
                                    top_k=int(self.hf_generation_defaults.get("top_k", 50)),   # not a key the reference reads (:705-715)
                                    repetition_penalty=float(samples.get("repetition_penalty", 1.0)),
                                    generator=samples.get("generator"), want_first_logits=want_first_logits,
                                    overlong="drop", num_beams=int(samples.get("num_beams", 1)),
                                    length_penalty=float(samples.get("length_penalty", 1.0)))
        # a prompt over max_pos costs ITS utterance, not the batch (the reference runs batch 1: inference/inference.py:370-373);
        # such rows come back pad-filled and are listed here for the caller (the CLI reports them as missing indices)
        self.last_dropped_rows = tuple(res.dropped)
        if res.dropped:
            logger.error("rows %s of this batch were not generated: prompt + new tokens exceed max_pos %d", list(res.dropped),
                         self.runtime.lm_cfg.max_pos)
        # host-side stage times of the last batch (launch of the speech encoders / prompt split + tokenisation / decoder incl.
        # the wait for the ids): the encoders run asynchronously under the tokenisation
        self.last_stage_seconds = {"speech_launch": t1 - t0, "segments": t2 - t1, "generate": time.perf_counter() - t2}
        self.batch_counter += 1
        return res

    def decode_ids(self, tokens) -> List[str]:
        return self.llama_tokenizer.batch_decode(tokens, skip_special_tokens=True)

    def generate_output(self, samples: Dict[str, Any]) -> List[str]:
        t0 = time.time()
        preds = self.decode_ids(self.generate_ids(samples).tokens)
        logger.debug("Generation took %.3f s", time.time() - t0)
        return preds

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "CustomSALMONN":
        return cls(**config)

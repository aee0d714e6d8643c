"""Input side of the hot path: prompt formatting, per-item processing and batch collation.

Mirrors the reference's ``SalmonProcessor`` (data/model_processors.py:476-874) for the classification
tasks: ``format_prompt`` reproduces ``_format_default_prompt`` (:742-776) character for character (the
prompt text fixes the token count S), ``process_inputs`` / ``collate_batch`` emit the batch dict of
SURVEY.md §8(a-0) (:616-681, :786-874) — with the boolean padding masks derived on demand from the lengths
(``CollatedBatch``) and the waveforms collated once, straight into the worker's shared-memory batch tensor.

MI355X-first difference: the log-mel (K1) moved onto the GPU, so by default the processor ships
``raw_wav`` (as float32 — the reference's ``torch.tensor(audio)`` keeps numpy float64) and NO
``spectrogram``; DataLoader workers never touch HIP.  Pass ``compute_spectrogram=True`` together with a
WhisperFeatureExtractor-compatible ``feature_extractor`` to reproduce the reference's CPU behaviour.
SQA items carry two audios (question + document) per query and per exemplar: ``_format_sqa_prompt`` (:697-740),
``_process_sqa_inputs`` (:510-614) and ``_collate_sqa_batch`` (:876-1009) are mirrored as well.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch

from .task_configs import DatasetType


_ARENA = None         # the collate destination of the current process while utils/batch_loader.collate_into(...) is active


def _batch_tensor(shape, dtype) -> torch.Tensor:
    """An uninitialised batch tensor; inside a DataLoader worker it is allocated in shared memory straight away (what
    ``default_collate`` does for stacked tensors), so handing the batch to the main process is a file-descriptor pass, not
    another 0.5 GB copy into shared memory."""
    if _ARENA is not None:                # utils/batch_loader.py: a preallocated, pinned, shared-memory slot of the batch loader
        t = _ARENA.take(shape, dtype)
        if t is not None:
            return t
    n = 1
    for d in shape:
        n *= int(d)
    if torch.utils.data.get_worker_info() is not None and n > 0:
        proto = torch.empty(0, dtype=dtype)
        return proto.new(proto._typed_storage()._new_shared(n)).resize_(*[int(d) for d in shape])
    return torch.empty(*[int(d) for d in shape], dtype=dtype)


def _pad_rows(rows: List[torch.Tensor], width: Optional[int] = None) -> torch.Tensor:
    """``pad_sequence(rows, batch_first=True)`` written once: every waveform is copied exactly once, into its row of the
    (shared-memory) batch tensor, and only the padding tail of a row is zero-filled."""
    width = max(int(r.numel()) for r in rows) if width is None else width
    out = _batch_tensor((len(rows), width), rows[0].dtype)
    for i, r in enumerate(rows):
        n = int(r.numel())
        out[i, :n] = r.reshape(-1)
        if n < width:
            out[i, n:] = 0
    return out


class CollatedBatch(dict):
    """The batch dict of SURVEY.md §8(a-0) with its boolean padding masks derived ON DEMAND.

    The reference's collate materialises ``padding_mask`` bool [B, L] (``data/model_processors.py:802``) and its model turns
    it straight back into lengths.  At 256 x 480 000 that is 123 MB per batch through worker shared memory, the pin thread
    and PCIe for a tensor nobody needs: the lengths travel as ``wav_lengths`` [B].  The masks are therefore not stored;
    ``batch["padding_mask"]`` (``"padding_mask" in batch``, ``batch.get("padding_mask")``) builds the reference's tensor from
    the lengths the first time a caller asks, on the device the waveforms live on."""

    _MASKS = {   # mask key -> (waveform key, lengths key)
        "padding_mask": ("raw_wav", "wav_lengths"),
        "example_padding_masks": ("example_wavs", "example_wav_lengths"),
        "question_padding_mask": ("question_raw_wav", "question_wav_lengths"),
        "document_padding_mask": ("document_raw_wav", "document_wav_lengths"),
        "example_question_padding_masks": ("example_question_wavs", "example_question_wav_lengths"),
        "example_document_padding_masks": ("example_document_wavs", "example_document_wav_lengths"),
    }

    def _derivable(self, key) -> bool:
        src = self._MASKS.get(key)
        return src is not None and dict.__contains__(self, src[0]) and dict.__contains__(self, src[1])

    def __contains__(self, key) -> bool:
        return dict.__contains__(self, key) or self._derivable(key)

    def __missing__(self, key):
        if not self._derivable(key):
            raise KeyError(key)
        wav_key, len_key = self._MASKS[key]
        wavs, lens = dict.__getitem__(self, wav_key), dict.__getitem__(self, len_key)
        lens = lens.to(wavs.device)
        mask = torch.arange(wavs.shape[-1], device=wavs.device).expand(wavs.shape) >= lens.unsqueeze(-1)
        self[key] = mask
        return mask

    def get(self, key, default=None):
        return self[key] if key in self else default

    def __reduce__(self):      # pickled (worker -> main process) and copied (pin_memory, device_prefetch) as this class
        return (type(self), (dict(self),))

    def copy(self):
        return type(self)(self)


def _is_sqa(dataset_type) -> bool:
    return dataset_type is not None and str(getattr(dataset_type, "value", dataset_type)) == "sqa"


class SalmonProcessor:
    def __init__(self, tokenizer, max_length: int = 128, feature_extractor=None, compute_spectrogram: bool = False):
        self.tokenizer = tokenizer
        self.max_length = max_length
        self.processor = feature_extractor
        self.compute_spectrogram = compute_spectrogram and feature_extractor is not None
        self.batch_counter = 0

    # ---- prompt text -----------------------------------------------------------------------------
    def format_prompt(self, template: str, text: str, examples: Optional[List[Dict]] = None,
                      input_mode: str = "speech_and_text", fewshot_mode: str = "text",
                      dataset_type: Optional[DatasetType] = None, **kwargs) -> str:
        if _is_sqa(dataset_type):
            return self._format_sqa_prompt(template, text, examples, input_mode, fewshot_mode, **kwargs)
        examples_text = ""
        if examples:
            if fewshot_mode == "speech":
                blocks = [f"<Speech><Example{i}></Speech>\nOutput: {ex.get('label', '')}" for i, ex in enumerate(examples)]
            else:
                blocks = [f"Text: {ex.get('text', '')}\nOutput: {ex.get('label', '')}" for ex in examples]
            examples_text = "\nHere are few examples to learn from:\n" + "\n\n".join(blocks) + "\n\n"
        if input_mode == "speech_and_text":
            input_section = f"<Speech><SpeechHere></Speech>\nTranscript: {text}"
        elif input_mode == "text_only":
            input_section = f"Text: {text}"
        else:
            input_section = "<Speech><SpeechHere></Speech>"
        return f"{template}\n{examples_text}Now analyze this input:\n{input_section}\nOutput:"

    def _format_sqa_prompt(self, template, text, examples, input_mode, fewshot_mode, **kwargs) -> str:
        """Character-exact restatement of :697-740, including the stray '>' before the exemplar question marker and the
        leading blanks of the reference's f-strings (they are part of the tokenised prompt)."""
        question = kwargs.get("question", "")
        examples_text = ""
        if examples:
            if fewshot_mode == "speech":
                blocks = [f"Document: <Speech><Document{i}></Speech>\nQuestion: ><Speech><Question{i}></Speech>\n"
                          f"Output: {ex.get('completion', '')}" for i, ex in enumerate(examples)]
            else:
                blocks = [f"Document: {ex.get('document', '')}\nQuestion: {ex.get('question', '')}\n"
                          f"Output: {ex.get('completion', '')}" for ex in examples]
            examples_text = "\nHere are few examples to learn from:\n" + "\n\n".join(blocks) + "\n\n"
        if input_mode == "speech_and_text":
            input_section = (f"Document: <Speech><Document></Speech>\nDocument text: {text}\n"
                             f"Question: <Speech><Question></Speech>\nQuestion text: {question}")
        elif input_mode == "text_only":
            input_section = f"\nDocument: {text}\nQuestion: {question}"
        else:
            input_section = "\nDocument: <Speech><Document></Speech>\n Question: <Speech><Question></Speech>"
        return f"{template}\n{examples_text} Now analyze this input:\n{input_section}\nOutput:"

    # ---- one item ----------------------------------------------------------------------------------
    def _audio(self, audio) -> Dict[str, Any]:
        arr = np.asarray(audio)
        if arr.dtype == np.float32 and not arr.flags.writeable:      # a view of a memory-mapped Arrow buffer: torch wants its own
            arr = arr.copy()
        wav = torch.as_tensor(arr, dtype=torch.float32).reshape(-1)
        out = {"raw_wav": wav, "wav_length": int(wav.numel()), "spectrogram": None}
        if self.compute_spectrogram:
            out["spectrogram"] = self.processor(np.asarray(audio), sampling_rate=16000, return_tensors="pt").input_features.squeeze(0)
        return out

    def process_inputs(self, data: Dict[str, Any], is_training: bool = False) -> Dict[str, Any]:
        if _is_sqa(data.get("dataset_type")):
            return self._process_sqa_inputs(data, is_training)
        input_mode = data.get("input_mode", "speech_only")
        tok = self.tokenizer(data.get("prompt", ""), padding="max_length", truncation=True, max_length=self.max_length,
                             return_tensors="pt")
        main = {"raw_wav": None, "wav_length": 0, "spectrogram": None}
        if data.get("audio") is not None and "speech" in input_mode:
            main = self._audio(data["audio"])
        examples = [self._audio(a) for a in (data.get("examples_audio") or [])]
        self.batch_counter += 1
        return {"input_ids": tok.input_ids, "attention_mask": tok.attention_mask, "spectrogram": main["spectrogram"],
                "raw_wav": main["raw_wav"], "wav_length": main["wav_length"], "examples_speech": examples,
                "num_examples": len(examples), "completion": data.get("completion", "")}

    def _process_sqa_inputs(self, data: Dict[str, Any], is_training: bool = False) -> Dict[str, Any]:
        input_mode = data.get("input_mode", "speech_only")
        tok = self.tokenizer(data.get("prompt", ""), padding="max_length", truncation=True, max_length=self.max_length,
                             return_tensors="pt")
        none = {"raw_wav": None, "wav_length": 0, "spectrogram": None}
        audio = data.get("audio") or {}
        speech = "speech" in input_mode
        q = self._audio(audio["question_audio"]) if speech and audio.get("question_audio") is not None else dict(none)
        d = self._audio(audio["document_audio"]) if speech and audio.get("document_audio") is not None else dict(none)
        examples = []
        for ex in (data.get("examples_audio") or []):
            examples.append({"question": self._audio(ex["question_audio"]) if ex.get("question_audio") is not None else dict(none),
                             "document": self._audio(ex["document_audio"]) if ex.get("document_audio") is not None else dict(none)})
        self.batch_counter += 1
        return {"input_ids": tok.input_ids, "attention_mask": tok.attention_mask,
                "question_spectrogram": q["spectrogram"], "question_raw_wav": q["raw_wav"], "question_wav_length": q["wav_length"],
                "document_spectrogram": d["spectrogram"], "document_raw_wav": d["raw_wav"], "document_wav_length": d["wav_length"],
                "examples_speech": examples, "num_examples": len(examples), "completion": data.get("completion", "")}

    # ---- batch -------------------------------------------------------------------------------------
    def collate_batch(self, items: List[Dict[str, Any]]) -> Dict[str, Any]:
        if _is_sqa(items[0].get("dataset_type")):
            return self._collate_sqa_batch(items)
        batch: Dict[str, Any] = CollatedBatch({
            "input_ids": torch.stack([it["input_ids"] for it in items]),
            "attention_mask": torch.stack([it["attention_mask"] for it in items]),
        })
        if all(it.get("raw_wav") is not None for it in items):
            batch["wav_lengths"] = torch.tensor([it["wav_length"] for it in items])
            batch["raw_wav"] = _pad_rows([it["raw_wav"] for it in items])       # "padding_mask": derived on demand (CollatedBatch)
            if all(it.get("spectrogram") is not None for it in items):
                batch["spectrogram"] = torch.stack([it["spectrogram"] for it in items])
        max_examples = max(it["num_examples"] for it in items)
        if max_examples > 0 and any(it.get("examples_speech") for it in items):
            max_len = max(ex["wav_length"] for it in items for ex in it["examples_speech"][:it["num_examples"]])
            have_spec = all(ex["spectrogram"] is not None for it in items for ex in it["examples_speech"])
            w = _batch_tensor((len(items), max_examples, max_len), torch.float32)
            l = torch.zeros(len(items), max_examples, dtype=torch.long)
            sp = torch.zeros(len(items), max_examples, 80, 3000) if have_spec else None
            for b, it in enumerate(items):
                row = it["examples_speech"][:it["num_examples"]]
                for e, ex in enumerate(row):
                    n = ex["wav_length"]
                    w[b, e, :n] = ex["raw_wav"]
                    w[b, e, n:] = 0
                    l[b, e] = n
                    if have_spec:
                        sp[b, e] = ex["spectrogram"]
                w[b, len(row):] = 0
            batch["example_wavs"], batch["example_wav_lengths"] = w, l           # "example_padding_masks": on demand
            if have_spec:
                batch["example_spectrograms"] = sp
        batch["num_examples"] = torch.tensor([it["num_examples"] for it in items])
        for key in ("prompt", "completion", "text", "dataset_type"):
            if key in items[0]:
                batch[key] = [it[key] for it in items]
        return batch


    def _collate_sqa_batch(self, items: List[Dict[str, Any]]) -> Dict[str, Any]:
        batch: Dict[str, Any] = CollatedBatch({"input_ids": torch.stack([it["input_ids"] for it in items]),
                                               "attention_mask": torch.stack([it["attention_mask"] for it in items])})
        if all(it.get("question_raw_wav") is not None and it.get("document_raw_wav") is not None for it in items):
            for side in ("question", "document"):
                batch[f"{side}_wav_lengths"] = torch.tensor([it[f"{side}_wav_length"] for it in items])
                batch[f"{side}_raw_wav"] = _pad_rows([it[f"{side}_raw_wav"] for it in items])   # masks: on demand
                if all(it.get(f"{side}_spectrogram") is not None for it in items):
                    batch[f"{side}_spectrogram"] = torch.stack([it[f"{side}_spectrogram"] for it in items])
        max_examples = max(it["num_examples"] for it in items)
        if max_examples > 0 and any(it.get("examples_speech") and all(ex[s]["raw_wav"] is not None for ex in it["examples_speech"]
                                                                        for s in ("question", "document")) for it in items):
            for side in ("question", "document"):
                exs = [[ex[side] for ex in it.get("examples_speech", [])[:it["num_examples"]]] for it in items]
                max_len = max(e["wav_length"] for row in exs for e in row)
                have_spec = all(e["spectrogram"] is not None for row in exs for e in row)
                w = torch.zeros(len(items), max_examples, max_len)
                l = torch.zeros(len(items), max_examples, dtype=torch.long)
                sp = torch.zeros(len(items), max_examples, 80, 3000) if have_spec else None
                for b, row in enumerate(exs):
                    for e, ex in enumerate(row):
                        n = ex["wav_length"]
                        w[b, e, :n], l[b, e] = ex["raw_wav"], n
                        if have_spec:
                            sp[b, e] = ex["spectrogram"]
                batch[f"example_{side}_wavs"], batch[f"example_{side}_wav_lengths"] = w, l
                if have_spec:
                    batch[f"example_{side}_spectrograms"] = sp
        batch["num_examples"] = torch.tensor([it["num_examples"] for it in items])
        for key in ("prompt", "completion", "text", "question", "dataset_type"):
            if key in items[0]:
                batch[key] = [it[key] for it in items]
        return batch


class QwenProcessor:
    """Mirror of the reference's ``QwenProcessor`` for the classification tasks (data/model_processors.py:153-223 inputs,
    :318-383 chat-template prompt, :428-472 collate).  ``processor`` is ``CustomQwen.input_processor``: its ``__call__``
    computes the 128-bin log-mel on the GPU, so build the DataLoader with ``num_workers=0`` for this model type (the
    reference's AutoProcessor runs the feature extractor on CPU workers)."""

    def __init__(self, processor, max_length: int = 512):
        self.processor = processor
        self.max_length = max_length

    def format_prompt(self, template: str, text: str, examples: Optional[List[Dict]] = None,
                      input_mode: str = "speech_only", fewshot_mode: str = "text", dataset_type=None, **kw) -> str:
        if _is_sqa(dataset_type):
            return self._format_sqa_prompt(template, text, examples, input_mode, fewshot_mode, **kw)
        user: List[Dict[str, Any]] = []
        if examples:
            user.append({"type": "text", "text": "Here are few examples to learn from:\n"})
            for ex in examples:
                if fewshot_mode == "speech":
                    user += [{"type": "audio", "audio_url": "dummy_url"}, {"type": "text", "text": f"Label: {ex.get('label', '')}\n"}]
                else:
                    user += [{"type": "text", "text": f"Text: {ex.get('text', '')}\n"},
                             {"type": "text", "text": f"Label: {ex.get('label', '')}\n"}]
        user.append({"type": "text", "text": "\nNow analyze this input:\n"})
        if input_mode in ("speech_only", "speech_and_text"):
            user.append({"type": "audio", "audio_url": "dummy_url"})
        if input_mode == "speech_and_text" and text:
            user.append({"type": "text", "text": text})
        conv = [{"role": "system", "content": template}, {"role": "user", "content": user}]
        return self.processor.apply_chat_template(conv, add_generation_prompt=True, tokenize=False)

    def _format_sqa_prompt(self, template, text, examples, input_mode, fewshot_mode, **kw) -> str:
        """SQA conversation (:240-316): per exemplar question audio, document audio, answer (speech few-shot) or three text
        items; then the query as question audio [+ text], document audio [+ text].  The reference reads the exemplar's
        ``answer`` key although the dataset fills ``completion`` (multi_task_dataset.py:262-270) — kept: answers render empty."""
        question = kw.get("question", "")
        user: List[Dict[str, Any]] = []
        if examples:
            user.append({"type": "text", "text": "Here are few examples to learn from:\n"})
            for i, ex in enumerate(examples):
                answer = ex.get("answer", "")
                if fewshot_mode == "speech":
                    user += [{"type": "audio", "audio_url": f"question_{i}"}, {"type": "audio", "audio_url": f"document_{i}"},
                             {"type": "text", "text": f"Answer: {answer}\n"}]
                else:
                    user += [{"type": "text", "text": f"Question: {ex.get('question', '')}\n"},
                             {"type": "text", "text": f"Document: {ex.get('document', '')}\n"},
                             {"type": "text", "text": f"Answer: {answer}\n"}]
        user.append({"type": "text", "text": "\nNow analyze this input:\n"})
        if input_mode in ("speech_and_text", "speech_only"):
            user.append({"type": "audio", "audio_url": "question"})
            if input_mode == "speech_and_text" and question:
                user.append({"type": "text", "text": f"Question text: {question}\n"})
            user.append({"type": "audio", "audio_url": "document"})
            if input_mode == "speech_and_text" and text:
                user.append({"type": "text", "text": f"Document text: {text}"})
        else:
            user += [{"type": "text", "text": f"Question: {question}\n"}, {"type": "text", "text": f"Document: {text}"}]
        conv = [{"role": "system", "content": template}, {"role": "user", "content": user}]
        return self.processor.apply_chat_template(conv, add_generation_prompt=True, tokenize=False)

    def process_inputs(self, data: Dict[str, Any], is_training: bool = False) -> Dict[str, Any]:
        text = data.get("prompt", "")
        if _is_sqa(data.get("dataset_type")):          # audio order (:85-100): exemplars (question, document)…, then the query's
            audios = []
            for ex in (data.get("examples_audio") or []):
                audios += [a for a in (ex.get("question_audio"), ex.get("document_audio")) if a is not None]
            main = data.get("audio") or {}
            audios += [a for a in (main.get("question_audio"), main.get("document_audio")) if a is not None]
        else:
            audios = list(data.get("examples_audio") or [])
            if data.get("audio") is not None:
                audios.append(data["audio"])
        full = text + (f"{data.get('completion', '')}{self.processor.tokenizer.eos_token}" if is_training else "")
        prompt_len = self.processor(text=text, audios=audios, return_tensors="pt", sampling_rate=16000).input_ids.shape[1]
        enc = self.processor(text=full, audios=audios, return_tensors="pt", sampling_rate=16000)
        out = {"input_ids": enc.input_ids.squeeze(0), "attention_mask": enc.attention_mask.squeeze(0), "prompt_length": prompt_len}
        if audios:
            out["input_features"] = enc.input_features
            out["feature_attention_mask"] = enc.feature_attention_mask
        return out

    def collate_batch(self, items: List[Dict[str, Any]]) -> Dict[str, Any]:
        S = max(it["input_ids"].numel() for it in items)
        pad = getattr(self.processor.tokenizer, "pad_token_id", 0) or 0
        ids = torch.full((len(items), S), pad, dtype=torch.long)
        att = torch.zeros(len(items), S, dtype=torch.long)
        for i, it in enumerate(items):     # right padding; the model strips it through attention_mask
            n = it["input_ids"].numel()
            ids[i, :n], att[i, :n] = it["input_ids"], it["attention_mask"]
        batch: Dict[str, Any] = {"input_ids": ids, "attention_mask": att,
                                 "prompt_length": torch.tensor([it["prompt_length"] for it in items])}
        if all("input_features" in it for it in items):
            batch["input_features"] = torch.cat([it["input_features"] for it in items])
            batch["feature_attention_mask"] = torch.cat([it["feature_attention_mask"] for it in items])
        for key in ("prompt", "completion", "text", "dataset_type"):
            if key in items[0]:
                batch[key] = [it[key] for it in items]
        return batch


def get_processor(model_type: str, processor=None, tokenizer=None, **kw):
    """Reference: data/model_processors.get_processor (:1012-1030)."""
    if model_type == "salmonn":
        return SalmonProcessor(tokenizer, feature_extractor=processor, **kw)
    if model_type == "qwen2":
        return QwenProcessor(processor, **kw)
    raise ValueError(f"Unsupported model type for the MI355X path: {model_type}")

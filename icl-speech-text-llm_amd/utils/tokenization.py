"""Tokenizers for the host side of the path.

The reference takes ``llama_tokenizer`` from the external SALMONN object (models/custom_salmon.py:109),
i.e. ``LlamaTokenizer.from_pretrained(llama_path, use_fast=False)`` plus a ``[PAD]`` special token
(pad id 32000 -> vocab 32001, SURVEY.md §7 "quirks").  ``load_llama_tokenizer`` does exactly that when
``llama_path`` is a local directory; offline (no checkpoint reachable) it falls back to ``ByteTokenizer``,
a dependency-free byte-level tokenizer that honours the small slice of the HF tokenizer call protocol the
glue uses (``__call__`` with ``padding`` / ``return_tensors="pt"`` / ``add_special_tokens``,
``batch_decode``, ``decode``, ``pad_token_id``, ``eos_token_id``, ``vocab_size``).
"""
from __future__ import annotations

import logging
import os
import re
import zlib
from typing import List, Sequence, Union

import torch

logger = logging.getLogger(__name__)


class Encoding(dict):
    """dict with attribute access and ``.to(device)`` (what the glue needs of a HF BatchEncoding)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def to(self, device):
        return Encoding({k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in self.items()})


class ByteTokenizer:
    """ids 0..2 = <unk>/<s>/</s>, 3..258 = bytes 0..255, then free ids up to vocab-2, pad = vocab-1."""

    bos_token_id, eos_token_id, unk_token_id = 1, 2, 0
    padding_side = "right"

    def __init__(self, vocab_size: int = 32001):
        assert vocab_size >= 260
        self._vocab = vocab_size
        self.pad_token_id = vocab_size - 1

    @property
    def vocab_size(self) -> int:  # HF: size without added tokens
        return self._vocab - 1

    def __len__(self) -> int:
        return self._vocab

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids = [b + 3 for b in text.encode("utf-8")]
        return ([self.bos_token_id] + ids) if add_special_tokens else ids

    def __call__(self, text: Union[str, Sequence[str]], padding=False, truncation=False, max_length=None,
                 return_tensors=None, add_special_tokens=True, return_attention_mask=True, **_):
        single = isinstance(text, str)
        rows = [self.encode(t, add_special_tokens) for t in ([text] if single else list(text))]
        if truncation and max_length:
            rows = [r[:max_length] for r in rows]
        width = max_length if padding == "max_length" and max_length else max((len(r) for r in rows), default=0)
        do_pad = padding in (True, "longest", "max_length")
        if do_pad or return_tensors == "pt":
            if not do_pad and len({len(r) for r in rows}) > 1:
                raise ValueError("cannot tensorise ragged rows without padding")
            mask = [[1] * len(r) + [0] * (width - len(r)) for r in rows]
            rows = [r + [self.pad_token_id] * (width - len(r)) for r in rows]
        else:
            mask = [[1] * len(r) for r in rows]
        if return_tensors == "pt":
            out = {"input_ids": torch.tensor(rows, dtype=torch.long).reshape(len(rows), width),
                   "attention_mask": torch.tensor(mask, dtype=torch.long).reshape(len(rows), width)}
        else:
            out = {"input_ids": rows[0] if single else rows, "attention_mask": mask[0] if single else mask}
        return Encoding(out)

    def decode(self, ids, skip_special_tokens: bool = False, **_) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        buf = bytearray()
        for t in ids:
            t = int(t)
            if 3 <= t < 259:
                buf.append(t - 3)
            elif not skip_special_tokens and t < 3:
                buf.extend(("<unk>", "<s>", "</s>")[t].encode())
            # ids >= 259 (unused range, pad) carry no text
        return buf.decode("utf-8", errors="replace")

    def batch_decode(self, batch, skip_special_tokens: bool = False, **kw) -> List[str]:
        if isinstance(batch, torch.Tensor):
            batch = batch.tolist()
        return [self.decode(row, skip_special_tokens=skip_special_tokens) for row in batch]


def _task_table_strings():
    """Every string of the shipped task tables (prompt templates, label sets, label mappings): the text the benchmark and the
    CLI tokenise over and over, and the labels whose first-token ids index the gathered logits."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "task_prompts.json")
    try:
        with open(path, encoding="utf-8") as f:
            table = json.load(f)
    except OSError:
        return
    stack = [table]
    while stack:
        o = stack.pop()
        if isinstance(o, str):
            yield o
        elif isinstance(o, dict):
            stack.extend(o.keys())
            stack.extend(o.values())
        elif isinstance(o, (list, tuple)):
            stack.extend(o)


class SubwordStandInTokenizer(ByteTokenizer):
    """Offline stand-in with the TOKEN COUNT of a sentencepiece Llama tokenizer (~3.9 characters per token on the
    reference's English prompts) rather than one token per byte: text is cut into space-prefixed word pieces
    of at most ``piece`` characters (5: 286 / 502 / 330 tokens on the VoxCeleb / HVB / VoxPopuli 5-shot prompts against the
    survey's 288 / 512 / 320).

    ``encode`` is a PURE FUNCTION of the text — the same ids in every process, rank and DataLoader worker (ranks exchange
    generated ids and label-logit columns, rank 0 decodes everybody's rows):
      * a fixed base vocabulary, built at construction from the shipped task tables (every piece of every prompt template
        and label, with and without a leading space, in sorted order) takes ids 259 ..; those always decode exactly;
      * any other piece takes ``base_end + crc32(piece) mod (free ids)``.  Two pieces may share such an id (a stand-in, not
        a vocabulary), so ``decode`` renders hashed ids as nothing — in EVERY process, whether or not it has encoded such a
        piece: decoding is a pure function of the ids too (only a random-weight model emits them; label words are base
        pieces).
    It exists so that benchmark prompts through the plugin have the prompt length the frozen workload definition assumes
    (376 positions for C2) when no tokenizer files are reachable; select it with ``llama_path="stand-in:subword"``."""

    def __init__(self, vocab_size: int = 32001, piece: int = 5):
        super().__init__(vocab_size)
        self.piece = piece
        self._split = re.compile(r"(?s).[^ \n]{0,%d}" % (piece - 1)).findall
        base = set()
        for text in _task_table_strings():
            base.update(self._split(text))
            base.update(self._split(" " + text))
        room = max(0, (self._vocab - 1 - 259) // 2)              # at most half of the free ids; the rest is the hashed range
        self._ids = {p: 259 + i for i, p in enumerate(sorted(base)[:room])}
        self._pieces = {t: p for p, t in self._ids.items()}
        self._hash0 = 259 + len(self._ids)
        self._hash_n = self._vocab - 1 - self._hash0             # ids _hash0 .. vocab-2 (vocab-1 = [PAD])

    def _id_of(self, p: str) -> int:
        t = self._ids.get(p)
        if t is None:
            if self._hash_n <= 0:
                return -1
            t = self._ids[p] = self._hash0 + zlib.crc32(p.encode("utf-8")) % self._hash_n      # memo only: never inverted
        return t

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        out: List[int] = [self.bos_token_id] if add_special_tokens else []
        # a piece = one character (of any kind) + up to piece-1 following characters that are neither space nor newline, so
        # it never crosses a word boundary; one regex pass instead of a per-character loop (128 prompts of ~1500 characters
        # per batch sit on the host path of the plugin benchmark)
        for p in self._split(text):
            t = self._id_of(p)
            if t < 0:                                             # no id range at all (tiny vocabularies): bytes
                out.extend(b + 3 for b in p.encode("utf-8"))
            else:
                out.append(t)
        return out

    def decode(self, ids, skip_special_tokens: bool = False, **_) -> str:
        if isinstance(ids, torch.Tensor):
            ids = ids.tolist()
        parts, buf = [], bytearray()
        for t in ids:
            t = int(t)
            if 3 <= t < 259:
                buf.append(t - 3)
                continue
            if buf:
                parts.append(buf.decode("utf-8", errors="replace")); buf = bytearray()
            if t in self._pieces:
                parts.append(self._pieces[t])
            elif not skip_special_tokens and t < 3:
                parts.append(("<unk>", "<s>", "</s>")[t])
        if buf:
            parts.append(buf.decode("utf-8", errors="replace"))
        return "".join(parts)


def load_llama_tokenizer(llama_path: str, vocab_size: int = 32001):
    if llama_path and os.path.isdir(llama_path):
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(llama_path, use_fast=False)
        tok.add_special_tokens({"pad_token": "[PAD]"})
        tok.padding_side = "right"
        return tok
    if llama_path == "stand-in:subword":
        return SubwordStandInTokenizer(vocab_size)
    if vocab_size >= 4096:
        # one token per BYTE makes the reference's 5-shot prompts 1100-2000 positions long (HVB overflows max_pos 2048);
        # the sub-word stand-in keeps them at the sentencepiece count (~3.9 characters per token)
        logger.warning("llama_path %r is not a local directory: using the sub-word stand-in tokenizer (no real vocabulary)", llama_path)
        return SubwordStandInTokenizer(vocab_size)
    logger.warning("llama_path %r is not a local directory: using the byte-level fallback tokenizer", llama_path)
    return ByteTokenizer(vocab_size)

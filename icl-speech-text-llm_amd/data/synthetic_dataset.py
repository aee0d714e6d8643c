"""Seeded synthetic stand-in for the reference's few-shot datasets.

The reference reads HF ``datasets`` folders from the authors' cluster (data/voxceleb_config.py:36-59) whose items
carry ``few_shot_examples`` chosen offline (data/multi_task_dataset.py:401-412).  None of that data exists here, so
this dataset emits items with the SAME processed-item schema (prompt built by the processor from the task's real
template, audio, exemplar texts/labels/audios, completion, text, dataset_type) from ``numpy.random.default_rng``:
audio N(0, 0.1^2) clipped to [-1,1].  Multi-task ordering follows the reference: sequential
concatenation by default, round-robin with ``interleave=True`` (data/multi_task_dataset.py:569-605).
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence

import numpy as np
from torch.utils.data import Dataset

from .task_configs import DatasetType, get_dataset_config

_WORDS = ("the committee agreed that people really enjoyed this wonderful evening although several members "
          "worried about rising costs and delays while others remained hopeful about future plans").split()


class SyntheticICLDataset(Dataset):
    def __init__(self, processor, dataset_types: Sequence[DatasetType], n_items: int = 32, num_examples: int = 5,
                 input_mode: str = "speech_only", fewshot_mode: str = "text", seed: int = 1234,
                 audio_seconds: float = 30.0, vary_length: bool = False, interleave: bool = False, text_chars: int = 75):
        self.processor = processor
        self.types = [DatasetType(t) for t in dataset_types]
        self.n_items, self.num_examples = n_items, num_examples
        self.input_mode, self.fewshot_mode = input_mode, fewshot_mode
        self.seed, self.audio_seconds, self.vary_length = seed, audio_seconds, vary_length
        self.interleave, self.text_chars = interleave, text_chars

    def __len__(self) -> int:
        return self.n_items * len(self.types)

    def _task_of(self, idx: int):
        if self.interleave:
            return self.types[idx % len(self.types)], idx // len(self.types)
        return self.types[idx // self.n_items], idx % self.n_items

    def _text(self, rng) -> str:
        out = ""
        while len(out) < self.text_chars:
            out += ("" if not out else " ") + _WORDS[int(rng.integers(len(_WORDS)))]
        return out[:self.text_chars]

    def _label(self, rng, cfg) -> str:
        labels = cfg.valid_labels
        if cfg.name == DatasetType.VOXCELEB:
            return labels[int(rng.integers(len(labels)))]
        k = int(rng.integers(1, 3))
        return ", ".join(sorted(rng.choice(labels, size=k, replace=False).tolist()))

    def _audio(self, rng) -> np.ndarray:
        secs = float(rng.uniform(2.0, 30.0)) if self.vary_length else self.audio_seconds
        n = int(round(secs * 16000))
        return np.clip(rng.normal(0.0, 0.1, n), -1.0, 1.0).astype(np.float32)

    def __getitem__(self, idx: int) -> Dict[str, Any]:
        dt, local = self._task_of(idx)
        cfg = get_dataset_config(dt)
        rng = np.random.default_rng(self.seed + idx)
        text, label = self._text(rng), self._label(rng, cfg)
        examples: List[Dict[str, Any]] = [{"text": self._text(rng), "label": self._label(rng, cfg)}
                                          for _ in range(self.num_examples)]
        audio = self._audio(rng) if "speech" in self.input_mode else None
        ex_audio = [self._audio(rng) for _ in examples] if self.fewshot_mode == "speech" else []
        prompt = self.processor.format_prompt(cfg.prompt_template, text, examples, input_mode=self.input_mode,
                                              fewshot_mode=self.fewshot_mode, dataset_type=dt)
        item = self.processor.process_inputs({"prompt": prompt, "audio": audio, "examples_audio": ex_audio,
                                              "completion": label, "input_mode": self.input_mode, "dataset_type": dt})
        item.update({"prompt": prompt, "completion": label, "text": text, "dataset_type": dt})
        return item

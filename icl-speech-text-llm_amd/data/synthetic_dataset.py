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


# ---------------------------------------------------------------------------------------------------------------------
# On-disk stand-ins with the reference's column schema (for ``--dataset_root`` runs, tests and golden generation)
# ---------------------------------------------------------------------------------------------------------------------
def _sentence(rng, n_chars: int) -> str:
    out = ""
    while len(out) < n_chars:
        out += ("" if not out else " ") + _WORDS[int(rng.integers(len(_WORDS)))]
    return out[:n_chars].strip()


def _clip(rng, seconds) -> Dict[str, Any]:
    n = int(round(float(rng.uniform(*seconds)) * 16000))
    return {"array": np.clip(rng.normal(0.0, 0.1, n), -1.0, 1.0).astype(np.float32).tolist(), "sampling_rate": 16000}


def _ner_row(rng, text: str, tags: List[str]):
    """A ``normalized_combined_ner`` cell: character spans of whole words of ``text`` with entity tags."""
    words, pos, spans = text.split(" "), 0, []
    for w in words:
        if w and rng.random() < 0.25:
            spans.append((str(rng.choice(tags)), pos, len(w)))
        pos += len(w) + 1
    return {"type": [s[0] for s in spans], "start": [s[1] for s in spans], "length": [s[2] for s in spans]}


def write_synthetic_hf_datasets(root: str, dataset_types: Sequence[DatasetType], n_items: int = 6, n_lookup: int = 8,
                                n_fewshot: int = 6, audio_seconds=(0.2, 0.6), seed: int = 0, splits=("test",)) -> None:
    """Writes, under ``root``, one HF ``datasets`` folder per (base task, split) plus its audio-lookup folder, named by the
    basenames of the reference's configured paths, with the columns the reference's item pipeline reads
    (data/multi_task_dataset.py:231-462): ``<text_key>``, ``<completion_key>``, ``audio{array,sampling_rate}``,
    ``few_shot_examples[{text,label,index}]`` and, in the lookup folder, ``index`` + ``audio`` (+ text / label columns for
    the tasks whose exemplars are sampled from it).  Audio is seeded noise stored as plain float lists (the image has no
    audio decoder, so the HF ``Audio`` feature is not used)."""
    import os
    import zlib
    import datasets
    from datasets import Dataset as HFDataset
    datasets.disable_progress_bars()
    from .task_configs import DatasetSplit, base_type_for_loading, set_dataset_root
    done = set()
    set_dataset_root(root)
    for dt in dataset_types:
        base = base_type_for_loading(dt)
        if base in done:
            continue
        done.add(base)
        cfg = get_dataset_config(base)
        name = base.name
        for split in splits:
            sp = DatasetSplit({"val": "validation"}.get(split, split))
            if sp not in cfg.paths or os.path.exists(cfg.get_path(sp)):
                continue                    # variants that share their base task's folders (e.g. MELD_GREEK) reuse them
            rng = np.random.default_rng(seed + zlib.crc32(os.path.basename(cfg.get_path(sp)).encode()))
            labels = list(cfg.label_mapping) if cfg.label_mapping else (cfg.valid_labels or [])   # folders hold base labels
            kind = cfg.completion_key

            def label_cell(text):
                if kind in ("sentiment", "sentiment_label", "emotion_label"):
                    return str(rng.choice(labels))
                if kind == "dialog_acts":
                    return rng.choice(labels, size=int(rng.integers(1, 4)), replace=False).tolist()
                if kind == "normalized_combined_ner":
                    return _ner_row(rng, text, labels)
                if kind == "answer_text":
                    s = float(np.round(rng.uniform(0, 20), 2))
                    return f"{s} {float(np.round(s + rng.uniform(0.3, 3), 2))}"
                if kind == "ne_spans":
                    return [{"label": str(rng.choice(["PER", "ORG", "LOC"])), "time_span": [float(np.round(t, 2)), float(np.round(t + 0.5, 2))]}
                            for t in rng.uniform(0, 5, int(rng.integers(0, 3)))]
                raise ValueError(f"no synthetic schema for {name} ({kind})")

            def example_label(text):        # the form exemplar labels take inside ``few_shot_examples``
                cell = label_cell(text)
                if kind == "normalized_combined_ner":
                    got = {}
                    for tag, start, length in zip(cell["type"], cell["start"], cell["length"]):
                        got.setdefault(tag, []).append(text[start:start + length])
                    return {tag: got.get(tag) for tag in labels}
                return cell

            lookup_rows = []
            for i in range(n_lookup):
                text = _sentence(rng, int(rng.integers(30, 70)))
                row = {"index": str(1000 + i), cfg.text_key: text, cfg.completion_key: label_cell(text)}
                if kind == "answer_text":
                    row.update({cfg.additional_text_keys["question"]: _sentence(rng, 25), "question_audio": _clip(rng, audio_seconds),
                                "document_audio": _clip(rng, audio_seconds), "unique_id": f"lk{i}"})
                else:
                    row["audio"] = _clip(rng, audio_seconds)
                lookup_rows.append(row)
            rows = []
            for i in range(n_items):
                text = _sentence(rng, int(rng.integers(40, 76)))
                row = {cfg.text_key: text, cfg.completion_key: label_cell(text)}
                if kind == "answer_text":
                    row.update({cfg.additional_text_keys["question"]: _sentence(rng, 25), "question_audio": _clip(rng, audio_seconds),
                                "document_audio": _clip(rng, audio_seconds), "unique_id": f"it{i}", "question_id": f"q{i}",
                                "document_id": f"d{i}"})
                else:
                    row["audio"] = _clip(rng, audio_seconds)
                    if kind == "ne_spans":
                        row.update({"unique_id": f"it{i}", "speaker_id": f"s{i % 3}"})
                    shots = []
                    for j in rng.choice(n_lookup, size=min(n_fewshot, n_lookup), replace=False):
                        t = lookup_rows[int(j)][cfg.text_key]
                        shots.append({"text": t, "label": example_label(t), "index": lookup_rows[int(j)]["index"]})
                    row["few_shot_examples"] = shots
                rows.append(row)
            for path, table in ((cfg.get_path(sp), rows), (cfg.get_audio_lookup_path(sp), lookup_rows)):
                if path and not os.path.exists(path):
                    HFDataset.from_list(table).save_to_disk(path)


def write_voxceleb_folder_arrow(root: str, n_items: int, seconds: float = 30.0, seed: int = 1234, n_fewshot: int = 5,
                                value_type: str = "float32") -> str:
    """A VOXCELEB test folder (+ its audio-lookup folder) of ``n_items`` rows with ``seconds``-long clips, written through Arrow
    arrays instead of Python lists (``write_synthetic_hf_datasets`` spends ~0.2 s per 30 s clip boxing floats: fine for six-row
    test folders, not for the few hundred clips a throughput run cycles through).  Same columns as the reference's item
    pipeline reads (data/multi_task_dataset.py:231-462): ``normalized_text``, ``sentiment``, ``audio{array, sampling_rate}``,
    ``few_shot_examples[{text, label, index}]``; audio = N(0, 0.1^2) clipped to [-1, 1], ``default_rng(seed + i)``
    (SURVEY.md §8d).  Returns the dataset folder."""
    import os
    import pyarrow as pa
    import datasets
    from datasets import Dataset as HFDataset
    datasets.disable_progress_bars()
    from .task_configs import DatasetSplit, set_dataset_root
    set_dataset_root(root)
    cfg = get_dataset_config(DatasetType.VOXCELEB)
    path, lookup_path = cfg.get_path(DatasetSplit.TEST), cfg.get_audio_lookup_path(DatasetSplit.TEST)
    if os.path.exists(path) and (not lookup_path or os.path.exists(lookup_path)):
        return path
    labels = list(cfg.label_mapping) if cfg.label_mapping else list(cfg.valid_labels)
    rng = np.random.default_rng(seed)
    n_lookup = max(8, n_fewshot + 3)
    lookup = [{"index": str(1000 + i), cfg.text_key: _sentence(rng, 75), cfg.completion_key: str(rng.choice(labels))} for i in range(n_lookup)]

    def audio_column(n_rows, n_samples, first_seed):
        vals = np.empty(n_rows * n_samples, dtype=value_type)
        for i in range(n_rows):
            vals[i * n_samples:(i + 1) * n_samples] = np.clip(np.random.default_rng(first_seed + i).normal(0.0, 0.1, n_samples), -1.0, 1.0)
        # 64-bit offsets: 512 clips of 30 s are 2.5e8 values, a few thousand would pass 2^31
        arr = pa.LargeListArray.from_arrays(pa.array(np.arange(n_rows + 1, dtype=np.int64) * n_samples), pa.array(vals))
        return pa.StructArray.from_arrays([arr, pa.array(np.full(n_rows, 16000, dtype=np.int64))], ["array", "sampling_rate"])

    n = int(round(seconds * 16000))
    texts = [_sentence(rng, 75) for _ in range(n_items)]
    shots = [[{"text": lookup[int(j)][cfg.text_key], "label": lookup[int(j)][cfg.completion_key], "index": lookup[int(j)]["index"]}
              for j in rng.choice(n_lookup, size=min(n_fewshot, n_lookup), replace=False)] for _ in range(n_items)]
    table = pa.table({cfg.text_key: pa.array(texts), cfg.completion_key: pa.array([str(rng.choice(labels)) for _ in range(n_items)]),
                      "audio": audio_column(n_items, n, seed), "few_shot_examples": pa.array(shots)})
    HFDataset(table).save_to_disk(path)
    if lookup_path:
        lk = pa.table({"index": pa.array([r["index"] for r in lookup]), cfg.text_key: pa.array([r[cfg.text_key] for r in lookup]),
                       cfg.completion_key: pa.array([r[cfg.completion_key] for r in lookup]),
                       "audio": audio_column(n_lookup, 16000, seed + 10 ** 6)})
        HFDataset(lk).save_to_disk(lookup_path)
    return path

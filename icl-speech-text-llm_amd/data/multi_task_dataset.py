"""Few-shot item pipeline in front of the hot path: dataset row → prompt text + audio + completion.

Mirrors the reference's ``data/multi_task_dataset.py``:
  ``BaseMultiTaskDataset``       :47-525   one task: exemplar selection, label formatting (lists, NER dicts, greek / swap
                                           mappings, SQA / NEL output formats), audio lookup by ``index``, prompt via the
                                           model processor, ``process_inputs``
  ``MultiTaskDataset`` (+Inference/Training)  :527-634   sequential concatenation, round-robin interleave (``idx % n_tasks``)
                                           or balanced sampling over per-task shuffled index tables
and ``data/{inference,training}_dataset.py``.  Pinned by tests/golden/dataset_items.json, captured from the reference's own
classes running over the same seeded on-disk datasets (``data/synthetic_dataset.write_synthetic_hf_datasets``).

Rows come from HF ``datasets`` folders (``load_from_disk``).  Audio columns are read as ``row["audio"]["array"]``; whatever
sequence type the folder yields is converted to float32 once, here, so the processor ships float32 ``raw_wav`` to the GPU
log-mel (the reference's ``torch.tensor(audio)`` keeps float64 — SURVEY.md §8 a-0).
"""
from __future__ import annotations

from ..utils.audio_io import decode_audio
from .arrow_audio import FastAudioRows

import logging
import random
import time
from typing import Any, Dict, List, Optional

import numpy as np
from torch.utils.data import Dataset

from .task_configs import DatasetSplit, DatasetType, get_dataset_config, get_swap_config, is_swap_type

logger = logging.getLogger(__name__)

_LOOKUP_SAMPLED = ("SQA", "VOXPOPULI_NEL", "MELD", "MELD_GREEK")     # exemplars drawn at random from the audio-lookup set
_NER_TASKS = ("VOXPOPULI", "VOXPOPULI_SWAP", "VOXPOPULI_GREEK")


def convert_ner_to_dict(text: str, ner_data: Dict) -> Dict[str, List[str]]:
    """``{type:[…], start:[…], length:[…]}`` → ``{tag: [phrases]}`` with empty phrases dropped (:19-44)."""
    out: Dict[str, List[str]] = {}
    for tag, start, length in zip(ner_data["type"], ner_data["start"], ner_data["length"]):
        phrase = text[start:start + length]
        if phrase.strip():
            out.setdefault(tag, []).append(phrase)
    return out


def _wave(x) -> Optional[np.ndarray]:
    """float32 waveform of an ``["array"]`` value (the reference's access path) or of a whole audio cell."""
    return decode_audio(x)


def _cell_wave(cell) -> Optional[np.ndarray]:
    """An audio CELL: decoded ``{"array", "sampling_rate"}`` or ``Audio(decode=False)``'s ``{"bytes", "path"}`` (utils/audio_io.py)."""
    return decode_audio(cell)


class BaseMultiTaskDataset(Dataset):
    def __init__(self, dataset_type: DatasetType, dataset, processor, input_mode: str = "speech_only",
                 fewshot_mode: str = "text", num_examples: int = 5, random_examples: bool = False,
                 split: DatasetSplit = DatasetSplit.TEST, model_type: str = "salmonn", run_name: str = "",
                 randomize_swap: bool = False, load_from_disk=None):
        self.dataset_type = DatasetType(dataset_type)
        self.dataset, self.processor = dataset, processor
        self.input_mode, self.fewshot_mode, self.num_examples = input_mode, fewshot_mode, num_examples
        self.random_examples = False                       # the reference overrides the argument (:87-88)
        self.split, self.model_type, self.run_name = split, model_type.lower(), run_name
        self.randomize_swap = randomize_swap
        self.config = get_dataset_config(self.dataset_type)
        self.is_swap_dataset = is_swap_type(self.dataset_type)
        if not self.is_swap_dataset:
            self.current_config = self.config
        # number-list audio columns are read zero-copy from the Arrow table (data/arrow_audio.py): dataset[idx] would box a 30 s
        # clip into 480 000 Python floats (~120 ms per row) only for the next line to turn them back into an array
        self._fast_rows = FastAudioRows.wrap(dataset)
        self._fast_lookup = None
        self.audio_lookup = None
        self.audio_index_map = None
        path = self.config.get_audio_lookup_path(self.split)
        if path:
            if load_from_disk is None:
                from datasets import load_from_disk
            t0 = time.time()
            self.audio_lookup = load_from_disk(path)
            self._fast_lookup = FastAudioRows.wrap(self.audio_lookup)
            if self.dataset_type.name not in _LOOKUP_SAMPLED:
                self.audio_index_map = {str(v): i for i, v in enumerate(self.audio_lookup["index"])}
            logger.info("Initialized audio lookup for %s in %.3fs", self.dataset_type, time.time() - t0)

    def __len__(self) -> int:
        return len(self.dataset)

    def _is_training(self) -> bool:
        return False

    # ---- labels ------------------------------------------------------------------------------------------------
    def _format_label(self, example_or_label, is_example: bool = True, current_mapping=None, text=None):
        label = example_or_label["label"] if is_example else example_or_label
        fmt = getattr(self.current_config, "output_format", None)
        if fmt == "timestamps_pair":
            return f"{label}"
        if fmt == "entity_timestamps":
            if not label:
                return "none"
            return "; ".join(f"{s['label']}: {s['time_span'][0]} {s['time_span'][1]}" for s in label)
        if self.dataset_type.name in _NER_TASKS and isinstance(label, dict):
            if not is_example:
                label = convert_ner_to_dict(text, label)
            kept = [k for k, v in label.items() if v]
            label = ", ".join(kept) if kept else "none"
        if isinstance(label, list):
            label = ", ".join(label)
        label = label.lower()
        mapping = current_mapping if current_mapping is not None else self.config.label_mapping
        if mapping and isinstance(label, str):
            if "," in label:
                label = ", ".join(mapping.get(p, p) for p in (q.strip().lower() for q in label.split(",")))
            else:
                label = mapping.get(label.lower(), label.lower())
        return label

    # ---- exemplars ---------------------------------------------------------------------------------------------
    def _select_examples(self, few_shot_examples):
        if self.random_examples:
            k = random.randint(0, self.num_examples)
            return random.sample(few_shot_examples, min(k, len(few_shot_examples))) if k > 0 else []
        return few_shot_examples[:self.num_examples]

    def _sample_lookup_indices(self) -> List[int]:
        total = len(self.audio_lookup)
        if self.random_examples:
            k = random.randint(0, self.num_examples)
            return random.sample(range(total), min(k, total)) if k > 0 else []
        return random.sample(range(total), min(self.num_examples, total))

    def _lookup_row(self, i: int):
        return self._fast_lookup.row(i) if self._fast_lookup is not None else self.audio_lookup[i]

    def _get_audio_by_index(self, index_str):
        if not index_str:
            return None
        if self.audio_lookup is None:
            logger.warning("Audio lookup not initialized for %s", self.dataset_type)
            return None
        try:
            i = self.audio_index_map.get(index_str)
            if i is None:
                logger.warning("No matching audio found for index %s", index_str)
                return None
            return self._lookup_row(i)["audio"]
        except Exception as e:
            logger.error("Error loading audio for index %s: %s", index_str, e)
            return None

    def _get_examples_audio(self, selected):
        if self.fewshot_mode != "speech":
            return None
        out = []
        for ex in selected:
            if "index" in ex:
                audio = self._get_audio_by_index(ex["index"])
                if audio is not None:
                    out.append(_cell_wave(audio))
        return out or None

    def _get_main_audio(self, item):
        if "speech" in self.input_mode and "audio" in item:
            return _cell_wave(item["audio"])
        return None

    @staticmethod
    def _get_audio_by_key(item, key):
        if key in item and item[key] is not None:
            return _cell_wave(item[key])
        return None

    # ---- items -------------------------------------------------------------------------------------------------
    def __getitem__(self, idx):
        if self.is_swap_dataset:
            self.current_config = get_swap_config(self.dataset_type, self.randomize_swap)
        item = self._fast_rows.row(idx) if self._fast_rows is not None else self.dataset[idx]
        if self.dataset_type.name == "SQA":
            return self._process_sqa_item(item, idx)
        return self._process_default_item(item, idx)

    def _process_default_item(self, item, idx):
        cfg = self.current_config
        examples: List[Dict[str, Any]] = []
        examples_audio: Optional[List[np.ndarray]] = []
        if self.dataset_type.name in _LOOKUP_SAMPLED and self.audio_lookup is not None and self.num_examples > 0:
            for i in self._sample_lookup_indices():
                ex = self._lookup_row(i)
                examples.append({"text": ex[cfg.text_key],
                                 "label": self._format_label(ex[cfg.completion_key], is_example=False,
                                                             current_mapping=cfg.label_mapping, text=ex[cfg.text_key])})
                if self.fewshot_mode == "speech" and ex.get("audio") is not None:
                    wave = _cell_wave(ex["audio"])           # decoded ONCE per exemplar (a 30 s WAV cell is ~1 MB of parsing)
                    if wave is not None:
                        examples_audio.append(wave)
        else:
            selected = self._select_examples(item.get("few_shot_examples", []))
            examples = [{"text": ex["text"],
                         "label": self._format_label(ex, is_example=True, current_mapping=cfg.label_mapping)}
                        for ex in selected]
            examples_audio = self._get_examples_audio(selected)
        prompt = self.processor.format_prompt(template=cfg.prompt_template, text=item[cfg.text_key], examples=examples,
                                              input_mode=self.input_mode, fewshot_mode=self.fewshot_mode,
                                              dataset_type=self.dataset_type)
        completion = self._format_label(item[cfg.completion_key], is_example=False, current_mapping=cfg.label_mapping,
                                        text=item[cfg.text_key])
        inputs = self.processor.process_inputs(
            data={"prompt": prompt, "fewshot_mode": self.fewshot_mode, "input_mode": self.input_mode,
                  "completion": completion, "audio": self._get_main_audio(item),
                  "examples_audio": examples_audio if examples_audio else None, "dataset_type": self.dataset_type},
            is_training=self._is_training())
        return {"prompt": prompt, "text": item[cfg.text_key], "completion": completion,
                "dataset_type": self.dataset_type, **inputs}

    def _process_sqa_item(self, item, idx):
        cfg = self.current_config
        q_key = cfg.additional_text_keys["question"]
        examples: List[Dict[str, Any]] = []
        examples_audio: Optional[List[Dict[str, Any]]] = None
        if self.audio_lookup is not None and self.num_examples > 0:
            examples_audio = []
            for i in self._sample_lookup_indices():
                ex = self._lookup_row(i)
                examples.append({"question": ex[q_key], "document": ex[cfg.text_key],
                                 "completion": self._format_label(ex[cfg.completion_key], is_example=False,
                                                                  current_mapping=cfg.label_mapping)})
                if self.fewshot_mode == "speech":
                    examples_audio.append({"question_audio": self._get_audio_by_key(ex, "question_audio"),
                                           "document_audio": self._get_audio_by_key(ex, "document_audio")})
        prompt = self.processor.format_prompt(template=cfg.prompt_template, text=item[cfg.text_key], question=item[q_key],
                                              examples=examples, input_mode=self.input_mode,
                                              fewshot_mode=self.fewshot_mode, dataset_type=self.dataset_type)
        inputs = self.processor.process_inputs(
            data={"prompt": prompt, "fewshot_mode": self.fewshot_mode, "input_mode": self.input_mode,
                  "completion": self._format_label(item[cfg.completion_key], is_example=False,
                                                   current_mapping=cfg.label_mapping),
                  "audio": {"question_audio": self._get_audio_by_key(item, "question_audio"),
                            "document_audio": self._get_audio_by_key(item, "document_audio")},
                  "examples_audio": examples_audio, "dataset_type": self.dataset_type},
            is_training=self._is_training())
        return {"prompt": prompt, "text": item[cfg.text_key], "question": item[q_key],
                "completion": item[cfg.completion_key], "dataset_type": self.dataset_type,
                "unique_id": item[cfg.additional_metadata_keys["unique_id"]], **inputs}


class InferenceDataset(BaseMultiTaskDataset):
    """data/inference_dataset.py:8-54 — always the TEST split, exemplars never randomised."""

    def __init__(self, dataset_type, dataset, processor, input_mode="speech_only", fewshot_mode="text", num_examples=5,
                 random_examples=False, model_type="salmonn", randomize_swap=False, **kw):
        super().__init__(dataset_type, dataset, processor, input_mode=input_mode, fewshot_mode=fewshot_mode,
                         num_examples=num_examples, random_examples=random_examples, split=DatasetSplit.TEST,
                         model_type=model_type, randomize_swap=randomize_swap, **kw)


class TrainingDataset(BaseMultiTaskDataset):
    """data/training_dataset.py:8-57 — TRAIN split, ``process_inputs(is_training=True)`` (training itself is out of scope)."""

    def __init__(self, dataset_type, dataset, processor, input_mode="speech_only", fewshot_mode="text", num_examples=5,
                 random_examples=True, model_type="salmonn", randomize_swap=True, **kw):
        super().__init__(dataset_type, dataset, processor, input_mode=input_mode, fewshot_mode=fewshot_mode,
                         num_examples=num_examples, random_examples=random_examples, split=DatasetSplit.TRAIN,
                         model_type=model_type, randomize_swap=randomize_swap, **kw)

    def _is_training(self) -> bool:
        return True


class MultiTaskDataset(Dataset):
    """Several tasks as one dataset (:527-617).  ``balance``: every task is tiled to the largest task's size; ``interleave``:
    item ``idx`` belongs to task ``idx % n_tasks``; both walk per-task index tables shuffled with ``np.random``.  Neither:
    tasks are concatenated in the given order (the inference default)."""

    def __init__(self, datasets: Dict[DatasetType, BaseMultiTaskDataset], processor, balance_datasets: bool = True,
                 interleave: bool = True):
        self.datasets, self.processor = datasets, processor
        self.dataset_types = list(datasets.keys())
        self.balance_datasets, self.interleave = balance_datasets, interleave
        self.dataset_sizes = {dt: len(ds) for dt, ds in datasets.items()}
        if balance_datasets:
            self.max_size = max(self.dataset_sizes.values())
            self.total_size = self.max_size * len(self.dataset_types)
            self.dataset_indices = {}
            for dt in self.dataset_types:
                size = self.dataset_sizes[dt]
                self.dataset_indices[dt] = np.tile(np.arange(size), -(-self.max_size // size))[:self.max_size]
                np.random.shuffle(self.dataset_indices[dt])
        elif interleave:
            self.max_size = max(self.dataset_sizes.values())
            self.total_size = sum(self.dataset_sizes.values())
            self.dataset_indices = {}
            for dt in self.dataset_types:
                self.dataset_indices[dt] = np.arange(self.dataset_sizes[dt])
                np.random.shuffle(self.dataset_indices[dt])
        else:
            self.total_size = sum(self.dataset_sizes.values())
            self.index_mapping = [(dt, i) for dt in self.dataset_types for i in range(self.dataset_sizes[dt])]

    def __len__(self) -> int:
        return self.total_size

    def locate(self, idx: int):
        """(task, row) that ``self[idx]`` reads."""
        if self.balance_datasets or self.interleave:
            dt = self.dataset_types[idx % len(self.dataset_types)]
            table = self.dataset_indices[dt]
            local = idx // len(self.dataset_types)
            return dt, int(table[local % (self.max_size if self.balance_datasets else len(table))])
        dt, local = self.index_mapping[idx]
        return dt, int(local)

    def __getitem__(self, idx):
        dt, row = self.locate(idx)
        item = self.datasets[dt][row]
        if "dataset_type" not in item:
            item["dataset_type"] = dt
        return item

    def on_epoch_end(self):
        if self.balance_datasets or self.interleave:
            for dt in self.dataset_types:
                np.random.shuffle(self.dataset_indices[dt])


class MultiTaskTrainingDataset(MultiTaskDataset):
    def _is_training(self):
        return True


class MultiTaskInferenceDataset(MultiTaskDataset):
    def __init__(self, datasets, processor, balance_datasets: bool = False, interleave: bool = False):
        super().__init__(datasets, processor, balance_datasets=balance_datasets, interleave=interleave)

    def _is_training(self):
        return False

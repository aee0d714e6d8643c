"""Task definitions that feed the hot path: dataset ids, splits, prompt templates, label sets, label mappings.

Mirrors the reference's ``data/base_config.py:5-66`` (``DatasetType``, ``DatasetSplit``, ``DatasetConfig``),
``data/master_config.py:36-104`` (``get_dataset_config``, ``get_swap_config``, ``apply_label_mapping``) and the swap
tables of ``data/{voxceleb,hvb,voxpopuli,meld_emotion}_config.py``.  The tables themselves (prompt templates, labels,
greek / swap mappings, dataset column names, folder names) are INPUT DATA of the path — they fix the prompt text and
hence S — and live verbatim in ``task_prompts.json``, exported from the reference by
``tests/golden/export_task_configs.py``.

Dataset folders: the reference hard-codes absolute paths on the authors' cluster.  They are kept as the default, and
``set_dataset_root(root)`` (CLI: ``--dataset_root``) re-roots every path to ``root/<basename>``.
"""
from __future__ import annotations

import json
import os
import random
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, List, Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(_HERE, "task_prompts.json")) as _f:
    _RAW = json.load(_f)

# data/base_config.py:5-36 — member names and values come from the exported table (VP_NEL and VOXPOPULI_NEL are distinct)
DatasetType = Enum("DatasetType", _RAW["dataset_types"], type=str, module=__name__)


class DatasetSplit(Enum):
    TRAIN = "train"
    VAL = "validation"
    TEST = "test"


_DATASET_ROOT: Optional[str] = None


def set_dataset_root(root: Optional[str]) -> None:
    global _DATASET_ROOT
    _DATASET_ROOT = root


def _reroot(path: Optional[str]) -> Optional[str]:
    if path is None or _DATASET_ROOT is None:
        return path
    return os.path.join(_DATASET_ROOT, os.path.basename(path.rstrip("/")))


@dataclass
class DatasetConfig:
    name: DatasetType
    paths: Dict[DatasetSplit, str]
    prompt_template: str
    valid_labels: Optional[List[str]]
    completion_key: str
    text_key: str
    audio_lookup_paths: Optional[Dict[DatasetSplit, str]] = None
    label_mapping: Optional[Dict[str, str]] = None
    additional_text_keys: Optional[Dict[str, str]] = None
    additional_audio_keys: Optional[Dict[str, str]] = None
    additional_metadata_keys: Optional[Dict[str, Any]] = None
    output_format: Optional[str] = None

    def get_path(self, split: DatasetSplit) -> str:
        return _reroot(self.paths[split])          # KeyError for a split the task lacks, as in the reference

    def get_audio_lookup_path(self, split: DatasetSplit) -> Optional[str]:
        if self.audio_lookup_paths:
            return _reroot(self.audio_lookup_paths.get(split))
        return None


def _build(d: Dict[str, Any]) -> DatasetConfig:
    return DatasetConfig(
        name=DatasetType(d["name"]),
        paths={DatasetSplit(k): v for k, v in (d.get("paths") or {}).items()},
        prompt_template=d["prompt_template"], valid_labels=d["valid_labels"], completion_key=d["completion_key"],
        text_key=d["text_key"],
        audio_lookup_paths=({DatasetSplit(k): v for k, v in d["audio_lookup_paths"].items()}
                            if d.get("audio_lookup_paths") else None),
        label_mapping=d.get("label_mapping"), additional_text_keys=d.get("additional_text_keys"),
        additional_metadata_keys=d.get("additional_metadata_keys"), output_format=d.get("output_format"))


DATASET_CONFIGS: Dict[DatasetType, DatasetConfig] = {DatasetType(k): _build(v) for k, v in _RAW["configs"].items()}
_SWAP_CONFIGS: Dict[DatasetType, List[DatasetConfig]] = {DatasetType(k): [_build(c) for c in v]
                                                         for k, v in _RAW["swap_configs"].items()}


def get_dataset_config(dataset_type) -> Optional[DatasetConfig]:
    """data/master_config.py:56-58 — ``None`` for a type without a table entry (e.g. VOXPOPULI_NEL)."""
    try:
        return DATASET_CONFIGS.get(DatasetType(dataset_type))
    except ValueError:
        return None


def get_swap_config(dataset_type, randomize: bool = False) -> DatasetConfig:
    """data/master_config.py:60-71: entry 1 of the family's table, or a ``random.choice`` of it."""
    family = _SWAP_CONFIGS.get(DatasetType(dataset_type))
    if family is None:
        raise ValueError(f"No swap config available for dataset type: {dataset_type}")
    return random.choice(family) if randomize else family[1]


_SWAP_TYPES = ("VOXCELEB_SWAP", "HVB_SWAP", "VOXPOPULI_SWAP", "MELD_EMOTION_SWAP")


def is_swap_type(dataset_type) -> bool:
    return DatasetType(dataset_type).name in _SWAP_TYPES


def base_type_for_loading(dataset_type) -> DatasetType:
    """utils/data_utils.py:50-64 — which task's folders a greek / swap variant is read from."""
    n = DatasetType(dataset_type).name
    for base in ("MELD_EMOTION", "VOXCELEB", "HVB"):
        if n in (base + "_GREEK", base + "_SWAP"):
            return DatasetType[base]
    if n in ("VOXPOPULI_GREEK", "VOXPOPULI_SWAP"):
        return DatasetType["VOXPOPULI"]
    return DatasetType(dataset_type)


def apply_label_mapping(examples: List[Dict], label_mapping: Dict[str, str]) -> List[Dict]:
    """data/master_config.py:73-96: first matching label column of each example is mapped in place."""
    for ex in examples:
        for key in ("sentiment", "sentiment_label", "emotion_label"):
            if key in ex:
                ex[key] = label_mapping.get(ex[key], ex[key])
                break
        else:
            if "dialog_acts" in ex:
                ex["dialog_acts"] = ",".join(label_mapping.get(a.strip(), a.strip()) for a in ex["dialog_acts"].split(","))
            elif "normalized_combined_ner" in ex:
                v = ex["normalized_combined_ner"]
                if isinstance(v, str) and v in label_mapping:
                    ex["normalized_combined_ner"] = label_mapping[v]
    return examples


def parse_dataset_types(arg: str) -> List[DatasetType]:
    """``--dataset_type a-b-c`` is hyphen separated (inference/inference.py:121-126)."""
    return [DatasetType(p.strip()) for p in arg.split("-")] if "-" in arg else [DatasetType(arg)]

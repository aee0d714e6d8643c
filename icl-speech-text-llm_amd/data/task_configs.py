"""Task definitions that feed the hot path: dataset ids, prompt templates, label sets.

Mirrors the slice of the reference's ``data/base_config.py:5-66`` / ``data/master_config.py`` /
``data/{voxceleb,hvb,voxpopuli}_config.py`` that the inference path reads (``prompt_template``,
``valid_labels``, ``completion_key``, ``text_key``).  The template strings are INPUT DATA of the
path (they determine the prompt length S), kept verbatim in ``task_prompts.json``; dataset paths on
the authors' cluster, the greek/swap label variants and the SQA / NEL / MELD tasks are out of scope
(SURVEY.md §2.1 #10, §8f-2).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from enum import Enum
from typing import Dict, List, Optional


class DatasetType(str, Enum):
    VOXCELEB = "voxceleb"
    HVB = "hvb"
    VOXPOPULI = "voxpopuli"


@dataclass(frozen=True)
class DatasetConfig:
    name: DatasetType
    prompt_template: str
    valid_labels: Optional[List[str]]
    completion_key: str
    text_key: str
    label_mapping: Optional[Dict[str, str]] = None


def _load() -> Dict[DatasetType, DatasetConfig]:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "task_prompts.json")) as f:
        raw = json.load(f)
    return {DatasetType(k): DatasetConfig(name=DatasetType(k), **v) for k, v in raw.items()}


_CONFIGS = _load()


def get_dataset_config(dataset_type) -> DatasetConfig:
    """Reference: data/master_config.get_dataset_config."""
    return _CONFIGS[DatasetType(dataset_type)]


def parse_dataset_types(arg: str) -> List[DatasetType]:
    """``--dataset_type a-b-c`` is hyphen separated (inference/inference.py:121-126)."""
    return [DatasetType(p.strip()) for p in arg.split("-")] if "-" in arg else [DatasetType(arg)]

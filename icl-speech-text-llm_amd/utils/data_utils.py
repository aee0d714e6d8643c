"""Dataset folder loading for the CLI — mirror of the reference's ``utils/data_utils.load_dataset`` (:22-91): a task's
split is an HF ``datasets`` folder read with ``load_from_disk``; greek / swap variants read their base task's folders;
loaded datasets are cached per (task, split)."""
from __future__ import annotations

import logging
import os
import time
from typing import Dict

from ..data.task_configs import DatasetSplit, DatasetType, base_type_for_loading, get_dataset_config

logger = logging.getLogger(__name__)

_DATASET_CACHE: Dict[str, object] = {}


def load_dataset(dataset_type: DatasetType, split="train", use_cache: bool = True):
    if isinstance(split, str):
        split = DatasetSplit({"val": "validation"}.get(split, split))     # the CLI says "val", the enum "validation"
    dataset_type = DatasetType(dataset_type)
    key = f"{dataset_type.value}_{split.value}"
    if use_cache and key in _DATASET_CACHE:
        return _DATASET_CACHE[key]
    path = get_dataset_config(base_type_for_loading(dataset_type)).get_path(split)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Dataset file not found: {path}")
    from datasets import load_from_disk
    t0 = time.time()
    from .audio_io import undecoded_audio_columns
    data = undecoded_audio_columns(load_from_disk(path))     # Audio columns still decode lazily where a backend exists
    logger.info("Loaded %d examples from %s %s in %.2fs", len(data), dataset_type, split, time.time() - t0)
    if use_cache:
        _DATASET_CACHE[key] = data
    return data


def clear_dataset_cache() -> int:
    n = len(_DATASET_CACHE)
    _DATASET_CACHE.clear()
    return n

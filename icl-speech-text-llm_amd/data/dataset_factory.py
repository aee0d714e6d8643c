"""Dataset construction for the CLI — mirror of the reference's ``data/dataset_factory.py:13-364``:
one task → ``InferenceDataset`` / ``TrainingDataset``; a list of tasks → ``MultiTaskInferenceDataset`` /
``MultiTaskTrainingDataset`` over per-task datasets (a task whose dataset is missing or fails to build is skipped with a
log line; no task left → error).  Every failure surfaces as ``RuntimeError("Failed to create dataset: …")``."""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Union

from .model_processors import get_processor
from .multi_task_dataset import (InferenceDataset, MultiTaskInferenceDataset, MultiTaskTrainingDataset, TrainingDataset)
from .task_configs import DatasetType, get_dataset_config

logger = logging.getLogger(__name__)

_INPUT_MODES = ("speech_only", "text_only", "speech_and_text")
_FEWSHOT_MODES = ("text", "speech")


class DatasetFactory:
    @staticmethod
    def create_dataset(dataset_type: Union[DatasetType, List[DatasetType]], dataset, processor, is_training: bool = False,
                       input_mode: str = "speech_only", fewshot_mode: str = "text", num_examples: int = 5,
                       random_examples: Optional[bool] = None, model_type: str = "salmonn", run_name: str = "",
                       randomize_swap: bool = False, balance_datasets: bool = False, interleave: bool = True):
        try:
            if input_mode not in _INPUT_MODES:
                raise ValueError(f"Invalid input_mode: {input_mode}. Must be one of: speech_only, text_only, speech_and_text")
            if fewshot_mode not in _FEWSHOT_MODES:
                raise ValueError(f"Invalid fewshot_mode: {fewshot_mode}. Must be one of: text, speech")
            if num_examples < 0:
                raise ValueError(f"Invalid num_examples: {num_examples}. Must be non-negative")
            if random_examples is None:
                random_examples = is_training
            common = dict(processor=processor, input_mode=input_mode, fewshot_mode=fewshot_mode, num_examples=num_examples,
                          random_examples=random_examples, model_type=model_type)
            one = TrainingDataset if is_training else InferenceDataset
            if not isinstance(dataset_type, list):
                return one(dataset_type=dataset_type, dataset=dataset, **common)
            per_task = {}
            for dt in dataset_type:
                rows = dataset.get(dt) if isinstance(dataset, dict) else dataset
                if rows is None:
                    logger.warning("No dataset provided for %s, skipping", dt)
                    continue
                try:
                    per_task[dt] = one(dataset_type=dt, dataset=rows, randomize_swap=randomize_swap, **common)
                except Exception as e:
                    logger.error("Error creating dataset for %s: %s", dt, e)
            if not per_task:
                raise ValueError("No valid datasets created for multi-task dataset")
            many = MultiTaskTrainingDataset if is_training else MultiTaskInferenceDataset
            return many(datasets=per_task, processor=processor, balance_datasets=balance_datasets, interleave=interleave)
        except Exception as e:
            logger.error("Error creating dataset: %s", e)
            raise RuntimeError(f"Failed to create dataset: {e}") from e

    @staticmethod
    def from_config(config: Dict[str, Any], datasets, processor=None):
        try:
            dataset_type = config.get("dataset_type")
            if not dataset_type:
                raise ValueError("dataset_type not specified in config")
            if isinstance(dataset_type, str):                      # comma separated here (:292-297), hyphens on the CLI
                dataset_type = ([DatasetType(p.strip()) for p in dataset_type.split(",")] if "," in dataset_type
                                else DatasetType(dataset_type))
            model_type = config.get("model_type", "salmonn")
            if processor is None:
                processor = get_processor(model_type, **config.get("processor_config", {}))
            return DatasetFactory.create_dataset(
                dataset_type=dataset_type, dataset=datasets, processor=processor,
                is_training=config.get("is_training", False), input_mode=config.get("input_mode", "speech_only"),
                fewshot_mode=config.get("fewshot_mode", "text"), num_examples=config.get("num_examples", 5),
                random_examples=config.get("random_examples"), model_type=model_type, run_name=config.get("run_name", ""))
        except Exception as e:
            logger.error("Error creating dataset from config: %s", e)
            raise RuntimeError(f"Failed to create dataset from config: {e}") from e

    @staticmethod
    def get_dataset_info(dataset_type: DatasetType) -> Dict[str, Any]:
        try:
            cfg = get_dataset_config(dataset_type)
            return {"name": DatasetType(dataset_type).value, "prompt_template": cfg.prompt_template,
                    "valid_labels": cfg.valid_labels, "completion_key": cfg.completion_key, "text_key": cfg.text_key,
                    "has_audio": cfg.audio_lookup_paths is not None}
        except Exception as e:
            logger.error("Error getting dataset info for %s: %s", dataset_type, e)
            return {"name": str(dataset_type), "error": str(e)}

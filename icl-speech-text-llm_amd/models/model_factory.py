"""Model factory — the drop-in boundary of the hot path (reference: models/model_factory.py:23-386).

``create_model(model_type, multi_task, task_configs, default_task, device, **model_kwargs)`` keeps the
reference's signature and error behaviour: unknown type -> ``ValueError`` re-raised as
``RuntimeError("Failed to create model: ...")`` (:60-61, :95-98).  ``multi_task=True`` wraps the model in
``MultiTaskModel`` (per-task generation knobs and prompt templates).  Unknown kwargs such as the ``use_cache`` that
``from_config`` forwards (:142) are tolerated by the model constructors.
"""
from __future__ import annotations

import logging
import time
import traceback
from typing import Any, Dict, Optional

import torch

logger = logging.getLogger(__name__)


class ModelFactory:
    @staticmethod
    def create_model(model_type: str, multi_task: bool = False, task_configs: Optional[Dict[str, Dict[str, Any]]] = None,
                     default_task: Optional[str] = None, device: Optional[torch.device] = None, **model_kwargs):
        try:
            model_type = model_type.lower()
            if model_type not in ("salmonn", "qwen2"):
                raise ValueError(f"Unknown model type: {model_type}")
            t0 = time.time()
            if multi_task:
                if not task_configs:
                    raise ValueError("task_configs required for multi-task models")
                from .multi_task_model import MultiTaskModel
                logger.info("Creating multi-task %s model", model_type)
                if "device" not in model_kwargs:
                    model_kwargs = dict(model_kwargs, device=device)
                model = MultiTaskModel(model_type=model_type, task_configs=task_configs, default_task=default_task, **model_kwargs)
            elif model_type == "salmonn":
                logger.info("Creating single-task %s model", model_type)
                from .custom_salmon import CustomSALMONN
                model = CustomSALMONN(device=device, **model_kwargs) if "device" not in model_kwargs else CustomSALMONN(**model_kwargs)
            else:
                from .custom_qwen import CustomQwen
                model = CustomQwen(device=device, **model_kwargs) if "device" not in model_kwargs else CustomQwen(**model_kwargs)
            if device is not None:
                model = model.to(device)
            logger.info("Model created in %.2fs", time.time() - t0)
            return model
        except Exception as e:
            logger.error("Error creating model: %s", e)
            logger.debug(traceback.format_exc())
            raise RuntimeError(f"Failed to create model: {e}") from e

    @staticmethod
    def from_config(config: Dict[str, Any], device: Optional[torch.device] = None, use_cache: bool = True):
        try:
            model_type = config.get("model_type")
            if not model_type:
                raise ValueError("model_type not specified in config")
            multi_task = config.get("multi_task", False)
            if multi_task and not config.get("task_configs"):
                raise ValueError("task_configs required for multi-task models")
            return ModelFactory.create_model(model_type=model_type, multi_task=multi_task,
                                             task_configs=config.get("task_configs"), default_task=config.get("default_task"),
                                             use_cache=use_cache, device=device, **config.get("model_params", {}))
        except Exception as e:
            logger.error("Error creating model from config: %s", e)
            raise RuntimeError(f"Failed to create model from config: {e}") from e

    @staticmethod
    def optimize_for_inference(model, **_):
        """Reference: ``.eval()`` + optional ``torch.compile`` (:224-269).  The HIP path has nothing to trace:
        its kernels are already compiled and the decode loop is captured in a HIP graph by the runtime."""
        model.eval()
        return model

    @staticmethod
    def get_model_from_checkpoint(checkpoint_path: str, model_type: str, device=None, **model_kwargs):
        """Reference: :327-386 — build, ``torch.load``, then the 4-way key dispatch of inference/inference.py:157-177."""
        model = ModelFactory.create_model(model_type, device=device, **model_kwargs)
        ckpt = torch.load(checkpoint_path, map_location="cpu")
        load_finetuned_checkpoint(model, ckpt)
        return model


def load_finetuned_checkpoint(model, checkpoint: Dict[str, Any]) -> int:
    """inference/inference.py:157-177: 'model_state_dict' | 'state_dict' -> model; 'model' -> model.salmonn; else raw."""
    for key in ("model_state_dict", "state_dict"):
        if key in checkpoint:
            model.load_state_dict(checkpoint[key], strict=False)
            return len(checkpoint[key])
    if "model" in checkpoint:
        model.salmonn.load_state_dict(checkpoint["model"], strict=False)
        return len(checkpoint["model"])
    model.load_state_dict(checkpoint, strict=False)
    return len(checkpoint)

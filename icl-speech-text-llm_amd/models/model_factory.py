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
from typing import Any, Dict, List, Optional

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
    def clear_cache():
        """Reference :202-222: empties the (never populated) model cache and the allocator's cache; returns the number of
        models dropped.  This factory keeps no cache either: what is handed back is the HIP allocator's free blocks."""
        import gc
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        logger.info("Model cache cleared: 0 models")
        return 0

    @staticmethod
    def optimize_for_inference(model, device=None):
        """Reference :224-269: move to ``device``, ``.eval()``, optional ``torch.compile``; the model comes back whatever
        happens.  The HIP path has nothing to trace: its kernels are compiled ahead of time and the decode loop is captured
        in a HIP graph by the runtime."""
        try:
            if device is not None:
                model = model.to(device)
            model.eval()
        except Exception as e:
            logger.error("Error optimizing model for inference: %s", e)
        return model

    _MODEL_INFO = {
        "salmonn": ("SALMONN", "Speech Audio Language Music Open Neural Network", "Multimodal model for speech, audio, and text",
                    {"llama_path": "/path/to/llama", "whisper_path": "openai/whisper-large-v2", "beats_path": "microsoft/beats"}),
        "qwen2": ("Qwen2", "Qwen2 Audio", "Multimodal model for audio and text", {"model_path": "Qwen/Qwen2-Audio-7B-Instruct"}),
    }

    @staticmethod
    def get_model_info(model_type: str) -> Dict[str, Any]:
        """Reference :271-315: a description record per model type; unknown type -> ValueError."""
        try:
            name, full, desc, paths = ModelFactory._MODEL_INFO[model_type.lower()]
        except KeyError:
            raise ValueError(f"Unknown model type: {model_type.lower()}") from None
        return {"name": name, "full_name": full, "description": desc, "default_paths": dict(paths), "supports_lora": True,
                "supports_speech": True, "supports_audio": True}

    @staticmethod
    def get_available_models() -> List[str]:
        return list(ModelFactory._MODEL_INFO)

    @staticmethod
    def get_model_from_checkpoint(checkpoint_path: str, base_model_path: str, model_type: str, device=None):
        """Reference :327-386: build the model on ``base_model_path`` (``llama_path`` for SALMONN, ``model_path`` for Qwen2), load
        the checkpoint's ``"model"`` / ``"model_state_dict"`` / raw state dict with ``strict=False``; a missing file or any
        failure -> ``RuntimeError("Failed to load model from checkpoint: ...")``."""
        import os
        try:
            if not os.path.exists(checkpoint_path):
                raise FileNotFoundError(f"Checkpoint not found at {checkpoint_path}")
            if model_type == "salmonn":
                from .custom_salmon import CustomSALMONN
                model = CustomSALMONN(llama_path=base_model_path, device=device)
            else:
                from .custom_qwen import CustomQwen
                model = CustomQwen(model_path=base_model_path, device=device)
            ckpt = torch.load(checkpoint_path, map_location="cpu")
            sd = next((ckpt[k] for k in ("model", "model_state_dict") if k in ckpt), ckpt)
            missing, unexpected = model.load_state_dict(sd, strict=False)
            if missing:
                logger.warning("Missing keys when loading checkpoint: %s", list(missing)[:20])
            if unexpected:
                logger.warning("Unexpected keys when loading checkpoint: %s", list(unexpected)[:20])
            return model
        except Exception as e:
            logger.error("Error loading model from checkpoint: %s", e)
            logger.debug(traceback.format_exc())
            raise RuntimeError(f"Failed to load model from checkpoint: {e}") from e


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

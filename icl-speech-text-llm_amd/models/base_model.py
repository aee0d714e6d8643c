"""Plugin contract of the ICL framework, kept call-compatible with the reference's
``models/base_model.py:8-143``: ``forward`` / ``generate_output`` / ``get_speech_embeddings`` /
``from_config`` are abstract, ``maybe_autocast`` and the whole-state-dict checkpoint helpers are
concrete.  The MI355X path computes in bf16 inside libicl_hip regardless of ``use_fp16`` (there is no
torch autocast region to enter), so ``maybe_autocast`` is a null context kept for callers that use it.
"""
from __future__ import annotations

import abc
import logging
from contextlib import nullcontext
from typing import Any, Dict, List, Tuple

import torch
import torch.nn as nn


class BaseModel(nn.Module, abc.ABC):
    def __init__(self, device=None, use_fp16: bool = False):
        super().__init__()
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.use_fp16 = use_fp16
        self.batch_counter = 0
        logging.info("Initializing model on device: %s, FP16 flag: %s", self.device, self.use_fp16)

    @abc.abstractmethod
    def forward(self, samples: Dict[str, Any]) -> Dict[str, Any]:
        """Teacher-forced pass; returns at least {'loss', 'logits', 'labels'}."""

    @abc.abstractmethod
    def generate_output(self, samples: Dict[str, Any]) -> List[str]:
        """Greedy generation; returns one decoded string per batch row."""

    @abc.abstractmethod
    def get_speech_embeddings(self, samples: Dict[str, Any]) -> Tuple:
        """(speech_embeds, speech_atts, example_embeds, example_atts)."""

    def maybe_autocast(self):
        return nullcontext()

    @classmethod
    @abc.abstractmethod
    def from_config(cls, config: Dict[str, Any]) -> "BaseModel":
        """Build an instance from a kwargs dictionary."""

    def save_checkpoint(self, path: str, optimizer=None, scheduler=None, epoch=None, loss=None):
        ckpt = {"model": self.state_dict(), "config": {"use_fp16": self.use_fp16}}
        for key, obj in (("optimizer", optimizer), ("scheduler", scheduler)):
            if obj is not None:
                ckpt[key] = obj.state_dict()
        if epoch is not None:
            ckpt["epoch"] = epoch
        if loss is not None:
            ckpt["loss"] = loss
        torch.save(ckpt, path)
        logging.info("Model checkpoint saved to %s", path)

    @classmethod
    def load_checkpoint(cls, path: str, config: Dict[str, Any] = None, map_location=None):
        logging.info("Loading checkpoint from %s", path)
        ckpt = torch.load(path, map_location=map_location)
        if config is None:
            if "config" not in ckpt:
                raise ValueError("No config provided and no config found in checkpoint")
            config = ckpt["config"]
        model = cls.from_config(config)
        model.load_state_dict(ckpt["model"] if "model" in ckpt else ckpt)
        return model, ckpt

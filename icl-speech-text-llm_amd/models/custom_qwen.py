"""Placeholder for the Qwen2-Audio plugin (reference: models/custom_qwen.py:29-247).

The Qwen2-Audio path (128-mel Whisper-style tower + AvgPool + projector + Qwen2 LM with QKV bias) reuses the same
kernels but is not wired up yet; constructing it fails loudly rather than silently falling back to eager PyTorch."""
from __future__ import annotations

from .base_model import BaseModel


class CustomQwen(BaseModel):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("CustomQwen (Qwen2-Audio) is not implemented on the MI355X path yet (SURVEY.md §8 a7)")

    def forward(self, samples):
        raise NotImplementedError

    def generate_output(self, samples):
        raise NotImplementedError

    def get_speech_embeddings(self, samples):
        raise NotImplementedError

    @classmethod
    def from_config(cls, config):
        return cls(**config)

"""Inference defaults = constructor kwargs of the model plugins.

Same keys as the reference's config/inference_config.py:28-53 (``get_inference_config(model_type)`` ->
``{"model_args": {...}}``, forwarded verbatim to ``ModelFactory.create_model``), plus the knobs the
MI355X path adds.  Paths are honoured when they exist on disk; otherwise the model is built with seeded
synthetic weights of the named architecture (there is no network on the box).
"""
from __future__ import annotations

from typing import Any, Dict, Optional


def get_inference_config(model_type: str, dataset_type: Optional[Any] = None) -> Dict[str, Any]:
    base = {"num_workers": 2, "batch_size": 1,
            "generation_args": {"max_new_tokens": 10, "temperature": 0.7, "top_p": 0.9, "do_sample": True}}
    if model_type == "salmonn":
        model = {"model_args": {
            "llama_path": "lmsys/vicuna-13b-v1.1",
            "whisper_path": "openai/whisper-large-v2",
            "beats_path": "/data2/neeraja/neeraja/BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt",
            "lora": True, "lora_rank": 8, "lora_alpha": 32, "lora_dropout": 0.05, "max_txt_len": 128,
        }}
    elif model_type == "qwen2":
        model = {"model_args": {"model_path": "Qwen/Qwen2-Audio-7B-Instruct", "lora": True, "max_txt_len": 512,
                                "lora_alpha": 32, "lora_dropout": 0.05, "lora_rank": 8, "ckpt_path": ""}}
    else:
        raise ValueError(f"Unsupported model type: {model_type}")
    cfg = {**base, **model}
    if dataset_type is not None:
        from ..data.task_configs import get_dataset_config
        try:
            dc = get_dataset_config(dataset_type)
            cfg.update({"prompt_template": dc.prompt_template, "valid_labels": dc.valid_labels})
        except Exception:
            pass
    return cfg

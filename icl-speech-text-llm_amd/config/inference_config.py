"""Inference defaults = constructor kwargs of the model plugins.

Same keys as the reference's config/inference_config.py:28-53 (``get_inference_config(model_type)`` ->
``{"model_args": {...}}``, forwarded verbatim to ``ModelFactory.create_model``), plus the knobs the
MI355X path adds.  Paths are honoured when they exist on disk; otherwise the model is built with seeded
synthetic weights of the named architecture (there is no network on the box).
"""
from __future__ import annotations

from typing import Any, Dict, Optional


# dataset_type -> (prompt_template, valid_labels) as the reference's function returns them (:62-80; its callers pass no dataset
# type and read only "model_args", so these short templates are NOT the task prompts of data/master_config.py)
_DATASET_EXTRAS = {
    "voxceleb": ("Determine the sentiment of the following speech: {text}", ["positive", "negative", "neutral"]),
    "hvb": ("Identify the emotions in the following speech: {text}", ["happy", "sad", "angry", "surprised", "fearful", "disgusted"]),
    "voxpopuli": ("Classify the sentiment of the following speech: {text}", ["positive", "negative", "neutral"]),
}
_QWEN_CKPT = "/data2/neeraja/neeraja/code/SALMONN/results/trained_models/ft_20e8b_qwen2_speech_text_voxceleb/final_model.pt"


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
        # the authors' fine-tuned checkpoint (:49): honoured when the file exists, skipped otherwise (models/custom_qwen.py)
        model = {"model_args": {"model_path": "Qwen/Qwen2-Audio-7B-Instruct", "lora": True, "max_txt_len": 512,
                                "lora_alpha": 32, "lora_dropout": 0.05, "lora_rank": 8, "ckpt_path": _QWEN_CKPT}}
    else:
        raise ValueError(f"Unsupported model type: {model_type}")
    cfg = {**base, **model}
    if dataset_type is not None:
        name = str(getattr(dataset_type, "value", dataset_type))
        for family, (template, labels) in _DATASET_EXTRAS.items():      # the base dataset and its _swap / _greek label variants only
            if name in (family, family + "_swap", family + "_greek"):
                cfg.update({"prompt_template": template, "valid_labels": list(labels)})
    return cfg

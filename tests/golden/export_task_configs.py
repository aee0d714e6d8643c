#!/usr/bin/env python3
"""Exports the reference's task tables (prompt templates, label sets, label mappings, dataset keys and folder names) as
DATA into ``icl-speech-text-llm_amd/data/task_prompts.json``.

Run in the build container only (imports ``data.master_config`` from /root/reference).  The prompt templates are inputs
of the hot path — they must be character-identical for the plugin to be a drop-in — so they are exported, not re-typed.
What is exported per DatasetType (data/master_config.py:36-54, data/base_config.py:43-66):
    name, prompt_template, valid_labels, completion_key, text_key, label_mapping, additional_text_keys,
    additional_metadata_keys, output_format, paths / audio_lookup_paths (the authors' absolute cluster paths are kept as
    given; ``--dataset_root`` re-roots their basenames)
and per swap family the ordered list of swap configurations (data/*_config.py ``*_SWAP_CONFIGS``; index 1 is the one used
when ``randomize_swap`` is False).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")

from data.base_config import DatasetType  # noqa: E402
from data.master_config import DATASET_CONFIGS  # noqa: E402
import data.hvb_config as hvb  # noqa: E402
import data.meld_emotion_config as meld_emotion  # noqa: E402
import data.voxceleb_config as voxceleb  # noqa: E402
import data.voxpopuli_config as voxpopuli  # noqa: E402


def dump(cfg):
    return {
        "name": cfg.name.value,
        "prompt_template": cfg.prompt_template,
        "valid_labels": cfg.valid_labels,
        "completion_key": cfg.completion_key,
        "text_key": cfg.text_key,
        "label_mapping": cfg.label_mapping,
        "additional_text_keys": cfg.additional_text_keys,
        "additional_metadata_keys": cfg.additional_metadata_keys,
        "output_format": cfg.output_format,
        "paths": {s.value: p for s, p in (cfg.paths or {}).items()},
        "audio_lookup_paths": {s.value: p for s, p in (cfg.audio_lookup_paths or {}).items()} or None,
    }


out = {
    "dataset_types": {m.name: m.value for m in DatasetType},
    "configs": {k.value: dump(c) for k, c in DATASET_CONFIGS.items()},
    "swap_configs": {
        "voxceleb_swap": [dump(c) for c in voxceleb.VOXCELEB_SWAP_CONFIGS],
        "hvb_swap": [dump(c) for c in hvb.HVB_SWAP_CONFIGS],
        "voxpopuli_swap": [dump(c) for c in voxpopuli.VOXPOPULI_SWAP_CONFIGS],
        "meld_emotion_swap": [dump(c) for c in meld_emotion.MELD_EMOTION_SWAP_CONFIGS],
    },
}
path = os.path.join(ROOT, "icl-speech-text-llm_amd", "data", "task_prompts.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1, ensure_ascii=False)
print(path, os.path.getsize(path) // 1024, "KiB;", len(out["configs"]), "configs")

"""Per-task wrapper around one model — mirror of the reference's ``models/multi_task_model.py:7-162``.

``task_configs[task]`` may hold a ``prompt_template`` and generation knobs (``max_new_tokens``, ``num_beams``,
``do_sample``, ``temperature``); ``generate_output`` takes the task of the first batch row, makes it current and writes its
knobs into the batch before delegating (:130-149); ``forward`` swaps each row's template for its task's when the batch
carries a ``task`` column (:108-120) and tags the outputs with the current task.  Unlike the reference's plain class, this
wrapper forwards ``to`` / ``eval`` / attribute reads to the wrapped model, so ``ModelFactory.create_model(multi_task=True,
device=…)`` works (the reference's ``model.to(device)`` at model_factory.py:88-90 has no ``to`` to call).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional

logger = logging.getLogger(__name__)


class MultiTaskModel:
    def __init__(self, model_type: str, task_configs: Optional[Dict[str, Dict[str, Any]]] = None,
                 default_task: Optional[str] = None, *args, **kwargs):
        self.model_type = model_type.lower()
        if self.model_type == "salmonn":
            from .custom_salmon import CustomSALMONN
            self.model = CustomSALMONN(*args, **kwargs)
        elif self.model_type == "qwen2":
            from .custom_qwen import CustomQwen
            self.model = CustomQwen(*args, **kwargs)
        else:
            raise ValueError(f"Unknown model type: {model_type}")
        self.task_configs = task_configs or {}
        self.current_task = default_task
        self.task_prompt_templates = {t: c["prompt_template"] for t, c in self.task_configs.items() if "prompt_template" in c}
        logger.info("Initialized MultiTaskModel (%s) with %d tasks", model_type, len(self.task_configs))

    # ---- delegation ------------------------------------------------------------------------------------------
    def __getattr__(self, name):
        if name == "model":
            raise AttributeError(name)
        return getattr(self.model, name)

    def to(self, *a, **kw):
        self.model = self.model.to(*a, **kw)
        return self

    def eval(self):
        self.model.eval()
        return self

    # ---- tasks -----------------------------------------------------------------------------------------------
    def set_task(self, task_name: str) -> bool:
        if task_name in self.task_configs:
            self.current_task = task_name
            return True
        logger.warning("Task '%s' not found in configured tasks", task_name)
        return False

    def get_task_prompt_template(self, task_name: Optional[str] = None) -> str:
        task = task_name or self.current_task
        if task in self.task_prompt_templates:
            return self.task_prompt_templates[task]
        return self.model.prompt_template      # AttributeError when the model has none, as in the reference (:70)

    def forward(self, samples: Dict[str, Any]) -> Dict[str, Any]:
        batch_tasks = samples.get("task", [self.current_task] * len(samples["prompt"]))
        if "task" in samples and any(t is not None for t in samples["task"]):
            base = self.model.prompt_template
            samples["prompt"] = [self.task_prompt_templates[t] + p.split(base, 1)[-1]
                                 if t is not None and t in self.task_prompt_templates else p
                                 for p, t in zip(samples["prompt"], batch_tasks)]
        out = self.model.forward(samples)
        out["task"] = self.current_task
        return out

    def generate_output(self, samples: Dict[str, Any]) -> List[str]:
        task = samples.get("task", [self.current_task])[0]
        if task:
            self.set_task(task)
        if task and task in self.task_configs:
            cfg = self.task_configs[task]
            samples.update({"max_new_tokens": cfg.get("max_new_tokens", 10), "num_beams": cfg.get("num_beams", 1),
                            "do_sample": cfg.get("do_sample", False), "temperature": cfg.get("temperature", 0.8)})
        return self.model.generate_output(samples)

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "MultiTaskModel":
        config = dict(config)
        return cls(model_type=config.pop("model_type"), task_configs=config.pop("task_configs", {}),
                   default_task=config.pop("default_task", None), **config)

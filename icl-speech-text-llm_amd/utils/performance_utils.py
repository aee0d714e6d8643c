"""Throughput accounting with the reference's definition of "examples per second" (SURVEY.md §8 a9).

``PerformanceTracker`` mirrors utils/performance_utils.py:15-127 of the reference — same constructor, ``update(step_time,
batch_size, loss=None, token_count=None)``, ``get_summary()`` keys and string formats: the counter is fed once per batch with
(inference seconds, batch size) (inference/inference.py:368) and ``examples_per_second = total_examples / (now - start_time)``
(:109), i.e. wall time since construction — data loading included, model load excluded.  ``timer`` / ``time_function`` are the
small helpers of :131-177.  The MI355X number of record comes from bench.py; this class keeps the CLI's log lines and the
``performance`` dict comparable with the reference's.  Pinned by tests/golden/performance_tracker.json (fake clock).
"""
from __future__ import annotations

import functools
import logging
import time
from contextlib import contextmanager
from typing import Any, Callable, Dict, Optional

import numpy as np


class PerformanceTracker:
    def __init__(self, log_interval: int = 100, logger=None):
        self.log_interval = log_interval
        self.logger = logger or logging.getLogger(__name__)
        self.reset()

    def reset(self):
        self.step_times, self.batch_sizes, self.loss_values = [], [], []
        self.start_time = time.time()
        self.last_log_time = self.start_time
        self.total_examples = 0
        self.total_tokens = 0
        self.step_count = 0

    def update(self, step_time: float, batch_size: int, loss: Optional[float] = None, token_count: Optional[int] = None):
        self.step_times.append(step_time)
        self.batch_sizes.append(batch_size)
        if loss is not None:
            self.loss_values.append(loss)
        self.total_examples += batch_size
        if token_count is not None:
            self.total_tokens += token_count
        self.step_count += 1
        if self.step_count % self.log_interval == 0:
            self.log_metrics()

    def log_metrics(self):
        now = time.time()
        elapsed = now - self.last_log_time
        recent = sum(self.batch_sizes[-self.log_interval:])
        metrics = {"avg_step_time": f"{np.mean(self.step_times[-self.log_interval:]):.4f}s",
                   "examples_per_second": f"{recent / elapsed:.2f}", "total_examples": self.total_examples}
        if self.loss_values:
            metrics["avg_loss"] = f"{np.mean(self.loss_values[-self.log_interval:]):.4f}"
        if self.total_tokens > 0:
            metrics["tokens_per_second"] = f"{recent * (self.total_tokens / self.total_examples) / elapsed:.2f}"
        self.logger.info(f"Performance metrics: {metrics}")
        self.last_log_time = now

    def get_summary(self) -> Dict[str, Any]:
        total = time.time() - self.start_time
        summary = {"total_time": f"{total:.2f}s", "total_examples": self.total_examples,
                   "avg_step_time": f"{np.mean(self.step_times):.4f}s",
                   "examples_per_second": f"{self.total_examples / total:.2f}", "step_count": self.step_count}
        if self.loss_values:
            summary["avg_loss"] = f"{np.mean(self.loss_values):.4f}"
        if self.total_tokens > 0:
            summary["tokens_per_second"] = f"{self.total_tokens / total:.2f}"
            summary["total_tokens"] = self.total_tokens
        return summary

    def log_summary(self):
        self.logger.info(f"Performance summary: {self.get_summary()}")


@contextmanager
def timer(name: str = None, logger=None):
    """``with timer("stage"):`` logs the wall time of the block (:131-149)."""
    log = logger or logging.getLogger(__name__)
    t0 = time.time()
    try:
        yield
    finally:
        elapsed = time.time() - t0
        log.info(f"{name} completed in {elapsed:.4f}s" if name else f"Operation completed in {elapsed:.4f}s")


def time_function(func: Callable) -> Callable:
    """Decorator form of ``timer`` (:152-177)."""
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        t0 = time.time()
        out = func(*args, **kwargs)
        logging.getLogger(__name__).info(f"Function '{func.__name__}' executed in {time.time() - t0:.4f} seconds")
        return out
    return wrapper

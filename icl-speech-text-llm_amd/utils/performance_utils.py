"""Throughput accounting with the reference's definition of "examples per second" (SURVEY.md §8 a9).

``PerformanceTracker`` keeps the CONTRACT of utils/performance_utils.py:15-127 of the reference — constructor, ``update(step_time,
batch_size, loss=None, token_count=None)``, ``get_summary()`` keys and string formats — over its own state (running sums and
bounded windows instead of per-step lists): the counter is fed once per batch with
(inference seconds, batch size) (inference/inference.py:368) and ``examples_per_second = total_examples / (now - start_time)``
(:109), i.e. wall time since construction — data loading included, model load excluded.  ``timer`` / ``time_function`` are the
small helpers of :131-177.  The MI355X number of record comes from bench.py; this class keeps the CLI's log lines and the
``performance`` dict comparable with the reference's.  Pinned by tests/golden/performance_tracker.json (fake clock).
"""
from __future__ import annotations

import functools
import logging
import math
import time
from collections import deque
from contextlib import contextmanager
from typing import Any, Callable, Deque, Dict, Optional


class _Window:
    """Sum over the last ``size`` values pushed (what a periodic log line averages)."""

    def __init__(self, size: int):
        self._buf: Deque[float] = deque(maxlen=max(int(size), 1))

    def push(self, value: float) -> None:
        self._buf.append(float(value))

    def mean(self) -> float:
        return math.fsum(self._buf) / len(self._buf) if self._buf else float("nan")

    def total(self) -> float:
        return math.fsum(self._buf)

    def __bool__(self) -> bool:
        return bool(self._buf)


class PerformanceTracker:
    """Running totals since construction plus sliding windows of ``log_interval`` entries for the periodic line; no per-step
    history is kept (a dataset run is millions of steps at 140 utterances/s).  One clock read per reset / log line / summary."""

    def __init__(self, log_interval: int = 100, logger=None):
        self.log_interval = log_interval
        self.logger = logger or logging.getLogger(__name__)
        self.reset()

    def reset(self):
        self.start_time = self._mark = time.time()        # public: callers may shift the origin (tests do)
        self.step_count = self.total_examples = self.total_tokens = 0
        self._seconds = self._loss_sum = 0.0
        self._loss_n = 0
        self._w_time, self._w_batch, self._w_loss = (_Window(self.log_interval) for _ in range(3))

    def update(self, step_time: float, batch_size: int, loss: Optional[float] = None, token_count: Optional[int] = None):
        self.step_count += 1
        self.total_examples += batch_size
        self.total_tokens += token_count or 0
        self._seconds += step_time
        self._w_time.push(step_time)
        self._w_batch.push(batch_size)
        if loss is not None:
            self._loss_sum += loss
            self._loss_n += 1
            self._w_loss.push(loss)
        if self.step_count % self.log_interval == 0:
            self.log_metrics()

    def log_metrics(self):
        now = time.time()
        rate = self._w_batch.total() / (now - self._mark)          # examples of the window over the time since the last line
        line = {"avg_step_time": f"{self._w_time.mean():.4f}s", "examples_per_second": f"{rate:.2f}",
                "total_examples": self.total_examples}
        if self._w_loss:
            line["avg_loss"] = f"{self._w_loss.mean():.4f}"
        if self.total_tokens > 0:
            line["tokens_per_second"] = f"{rate * self.total_tokens / self.total_examples:.2f}"
        self.logger.info(f"Performance metrics: {line}")
        self._mark = now

    def get_summary(self) -> Dict[str, Any]:
        wall = time.time() - self.start_time
        mean_step = self._seconds / self.step_count if self.step_count else float("nan")
        out: Dict[str, Any] = {"total_time": f"{wall:.2f}s", "total_examples": self.total_examples,
                               "avg_step_time": f"{mean_step:.4f}s",
                               "examples_per_second": f"{self.total_examples / wall:.2f}", "step_count": self.step_count}
        if self._loss_n:
            out["avg_loss"] = f"{self._loss_sum / self._loss_n:.4f}"
        if self.total_tokens > 0:
            out.update(tokens_per_second=f"{self.total_tokens / wall:.2f}", total_tokens=self.total_tokens)
        return out

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

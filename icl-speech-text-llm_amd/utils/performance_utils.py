"""Throughput accounting with the reference's definition of "examples per second".

``PerformanceTracker`` restates utils/performance_utils.py:15-127 of the reference: the counter is
fed once per batch with (inference seconds, batch size) (inference/inference.py:368) and
``examples_per_second = total_examples / (now - start_time)`` (:109) — wall time since construction,
i.e. including data loading, excluding model load.  The MI355X number of record comes from bench.py;
this class keeps the CLI's log lines comparable with the reference's.
"""
from __future__ import annotations

import logging
import time
from typing import Dict

logger = logging.getLogger(__name__)


class PerformanceTracker:
    def __init__(self, log_interval: int = 10):
        self.log_interval = log_interval
        self.reset()

    def reset(self):
        self.start_time = time.time()
        self.batch_times = []
        self.total_examples = 0
        self.total_batches = 0

    def update(self, batch_time: float, batch_size: int):
        self.batch_times.append(batch_time)
        self.total_examples += batch_size
        self.total_batches += 1
        if self.log_interval and self.total_batches % self.log_interval == 0:
            self.log_metrics()

    def get_summary(self) -> Dict[str, float]:
        total = time.time() - self.start_time
        n = max(len(self.batch_times), 1)
        return {
            "total_time": total,
            "total_batches": self.total_batches,
            "total_examples": self.total_examples,
            "avg_batch_time": sum(self.batch_times) / n,
            "examples_per_second": self.total_examples / total if total > 0 else 0.0,
            "batches_per_second": self.total_batches / total if total > 0 else 0.0,
        }

    def log_metrics(self):
        s = self.get_summary()
        logger.info("Performance: %.2f examples/s, avg batch %.4f s, %d examples", s["examples_per_second"],
                    s["avg_batch_time"], s["total_examples"])

    def log_summary(self):
        s = self.get_summary()
        logger.info("=== Performance Summary ===")
        for k, v in s.items():
            logger.info("%s: %s", k, f"{v:.4f}" if isinstance(v, float) else v)

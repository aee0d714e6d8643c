"""Dataset folder loading for the CLI — mirror of the reference's ``utils/data_utils.load_dataset`` (:22-91): a task's
split is an HF ``datasets`` folder read with ``load_from_disk``; greek / swap variants read their base task's folders;
loaded datasets are cached per (task, split)."""
from __future__ import annotations

import logging
import os
import time
from typing import Dict

from ..data.task_configs import DatasetSplit, DatasetType, base_type_for_loading, get_dataset_config

logger = logging.getLogger(__name__)

_DATASET_CACHE: Dict[str, object] = {}


def load_dataset(dataset_type: DatasetType, split="train", use_cache: bool = True):
    if isinstance(split, str):
        split = DatasetSplit({"val": "validation"}.get(split, split))     # the CLI says "val", the enum "validation"
    dataset_type = DatasetType(dataset_type)
    path = get_dataset_config(base_type_for_loading(dataset_type)).get_path(split)
    key = f"{dataset_type.value}_{split.value}|{path}"      # the folder is part of the key: --dataset_root may re-root a task between runs
    if use_cache and key in _DATASET_CACHE:
        return _DATASET_CACHE[key]
    if not os.path.exists(path):
        raise FileNotFoundError(f"Dataset file not found: {path}")
    from datasets import load_from_disk
    t0 = time.time()
    from .audio_io import undecoded_audio_columns
    data = undecoded_audio_columns(load_from_disk(path))     # Audio columns still decode lazily where a backend exists
    logger.info("Loaded %d examples from %s %s in %.2fs", len(data), dataset_type, split, time.time() - t0)
    if use_cache:
        _DATASET_CACHE[key] = data
    return data


def clear_dataset_cache() -> int:
    n = len(_DATASET_CACHE)
    _DATASET_CACHE.clear()
    return n


def device_prefetch(loader, device):
    """Iterate ``loader`` one batch AHEAD: the tensors of batch i+1 are copied to ``device`` on a side stream (truly asynchronous
    when the DataLoader pins its batches) while the caller's kernels for batch i run, instead of a blocking ``.to(device)``
    between two batches (246 MB of raw audio per 128 utterances).  Yields batch dicts whose tensors live on ``device``; the
    consumer's current stream waits for the copy of the batch it receives.  On a CPU device it is a plain pass-through."""
    import copy
    import torch
    dev = torch.device(device)
    if dev.type != "cuda":
        for batch in loader:
            yield batch
        return
    side = torch.cuda.Stream(device=dev)

    def ship(batch):
        with torch.cuda.stream(side):
            # only the floating-point payload (waveforms, spectrograms) goes to the device: lengths, masks and example counts
            # are read on the host by the model's prompt logic, and a device copy would turn each such read into a stream sync
            # in the middle of a batch (the encoders are running by then)
            out = copy.copy(batch)     # keeps the batch's own mapping type (CollatedBatch derives its padding masks on demand)
            out.update({k: v.to(dev, non_blocking=True) for k, v in batch.items()
                        if isinstance(v, torch.Tensor) and v.is_floating_point()})
        ev = torch.cuda.Event()
        ev.record(side)
        return out, ev, batch          # keep the pinned host batch alive until its copy has been waited for

    it = iter(loader)
    try:
        pending = ship(next(it))
    except StopIteration:
        return
    while pending is not None:
        cur, ev, _host = pending
        try:
            pending = ship(next(it))
        except StopIteration:
            pending = None
        torch.cuda.current_stream(dev).wait_event(ev)
        for v in cur.values():
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(torch.cuda.current_stream(dev))
        yield cur

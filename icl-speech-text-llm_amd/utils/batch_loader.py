"""Batch loader of the inference loop: worker processes collate straight into preallocated, pinned, shared-memory slots.

Where it sits: the reference iterates ``DataLoader(dataset, batch_size, shuffle=False, num_workers, pin_memory, collate_fn)``
(``inference/inference.py:259-266``) and moves each batch to the device inside the model.  At one utterance per 20 s that
loop costs nothing; at 140 utterances/s per GPU a batch is 0.5 GB of raw audio every 1.8 s, and the stock pipeline handles
those bytes four times on FRESH memory — the worker's ``pad_sequence`` output, its copy into a new shared-memory file, the pin
thread's copy into a new page-locked block, then the DMA — and first-touch page faults on half a gigabyte per batch are what
bound it (measured: 1.2 s of a worker's time per 256-utterance batch for the collate alone, 20 ms when the destination pages
already exist).  Here the destination exists ONCE:

* the main process allocates ``slots`` arenas as anonymous shared mappings before it forks the workers and page-locks them
  (``hipHostRegister``), so a worker's collate writes each waveform exactly once, into pages that are already mapped, and the
  H2D copy is a direct DMA from that same memory on a side stream while the previous batch's kernels run;
* a worker returns small tensors / strings through a queue and, for everything it placed in the arena, only
  ``(offset, shape, dtype)``; the main process views its own mapping of the slot;
* a slot goes back to the free list when the event recorded behind its H2D copy has completed; batches are handed out in
  dataset order whatever order the workers finish in; a worker exception is re-raised in the main process.

Workers never touch HIP.  ``num_workers=0`` runs the same code inline.  On a CPU device there is no copy: the yielded batch
aliases its slot until the next batch is requested (the iteration protocol of a DataLoader with ``pin_memory``).
"""
from __future__ import annotations

import collections
import contextlib
import logging
import os
import queue
import traceback
from typing import Any, Callable, Dict, List, Optional, Sequence

import torch

from ..data import model_processors as _mp

logger = logging.getLogger(__name__)

_ALIGN = 256


class CollateArena:
    """A bump allocator over one flat shared-memory byte buffer; ``model_processors._batch_tensor`` draws batch tensors from the
    active arena.  ``nbytes == 0``: a measuring arena (hands out nothing, records what was asked for)."""

    def __init__(self, nbytes: int, shared: bool = True):
        self.nbytes = int(nbytes)
        self.buffer = None
        self._map = None
        if self.nbytes:
            if shared:
                # an anonymous MAP_SHARED mapping: inherited by the forked workers at the same address, and not a file under
                # /dev/shm (whose size limit — 64 MB in a default container — eight ranks x ten half-gigabyte slots would hit)
                import mmap
                self._map = mmap.mmap(-1, self.nbytes, flags=mmap.MAP_SHARED | mmap.MAP_ANONYMOUS)
                self.buffer = torch.frombuffer(self._map, dtype=torch.uint8)
            else:
                self.buffer = torch.empty(self.nbytes, dtype=torch.uint8)
            self.buffer.zero_()                       # touch every page once, here, not under a timed batch
        self.used = 0
        self.asked = 0
        self.pinned = False

    def reset(self) -> None:
        self.used = self.asked = 0

    def take(self, shape: Sequence[int], dtype) -> Optional[torch.Tensor]:
        n = 1
        for d in shape:
            n *= int(d)
        size = n * torch.empty(0, dtype=dtype).element_size()
        off = -(-self.used // _ALIGN) * _ALIGN
        self.asked = -(-self.asked // _ALIGN) * _ALIGN + size
        if self.buffer is None or n == 0 or off + size > self.nbytes:
            return None
        self.used = off + size
        return self.buffer[off:off + size].view(dtype).view(*[int(d) for d in shape])

    def offset_of(self, t: torch.Tensor) -> Optional[int]:
        if self.buffer is None or not isinstance(t, torch.Tensor) or t.device.type != "cpu" or not t.is_contiguous():
            return None
        off = t.data_ptr() - self.buffer.data_ptr()
        return off if 0 <= off and off + t.numel() * t.element_size() <= self.nbytes and t.numel() else None

    def view(self, off: int, shape, dtype) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        size = n * torch.empty(0, dtype=dtype).element_size()
        return self.buffer[off:off + size].view(dtype).view(*shape)

    def pin(self) -> bool:
        """Page-lock the slot for direct DMA (main process, before the workers are forked).  False when it cannot be."""
        if self.buffer is None or self.pinned or not torch.cuda.is_available():
            return self.pinned
        try:
            rc = int(torch.cuda.cudart().cudaHostRegister(self.buffer.data_ptr(), self.nbytes, 1))
            self.pinned = rc == 0
        except Exception as e:       # an unpinned slot still works: the copy is staged by the driver
            logger.warning("could not page-lock a %d-byte collate slot: %s", self.nbytes, e)
        return self.pinned

    def unpin(self) -> None:
        if self.pinned:
            try:
                torch.cuda.cudart().cudaHostUnregister(self.buffer.data_ptr())
            except Exception:
                pass
            self.pinned = False


@contextlib.contextmanager
def collate_into(arena: Optional[CollateArena]):
    prev = _mp._ARENA
    _mp._ARENA = arena
    try:
        yield
    finally:
        _mp._ARENA = prev


def _pack(batch: Dict[str, Any], arena: CollateArena):
    """Batch -> (mapping type, payload): arena tensors become ('@', offset, shape, dtype)."""
    payload = {}
    for k, v in batch.items():
        off = arena.offset_of(v) if isinstance(v, torch.Tensor) else None
        payload[k] = ("@arena", off, tuple(v.shape), v.dtype) if off is not None else v
    return type(batch), payload


def _unpack(kind, payload: Dict[str, Any], arena: CollateArena):
    out = {}
    for k, v in payload.items():
        if isinstance(v, tuple) and len(v) == 4 and v[0] == "@arena":
            out[k] = arena.view(v[1], v[2], v[3])
        else:
            out[k] = v
    try:
        return kind(out)
    except Exception:
        return out


def _worker_main(wid: int, dataset, collate_fn, arenas: List[CollateArena], task_q, result_q) -> None:
    torch.set_num_threads(1)                 # the parallelism is across workers; an intra-op pool per worker oversubscribes
    while True:
        task = task_q.get()
        if task is None:
            return
        batch_no, idxs, slot = task
        try:
            arena = arenas[slot]
            arena.reset()
            with collate_into(arena):
                batch = collate_fn([dataset[i] for i in idxs])
            kind, payload = _pack(batch, arena)
            result_q.put((batch_no, slot, kind, payload, None))
        except Exception as e:               # surfaced in the main process with the worker's traceback
            result_q.put((batch_no, slot, None, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


class ArenaBatchLoader:
    """``for batch in ArenaBatchLoader(dataset, batch_size, collate_fn, num_workers, device)``: batches in dataset order (no
    shuffling: the reference's inference loader never shuffles), tensors already on ``device`` when it is a GPU.

    ``slot_bytes``: arena size; default = what the first batch asks for, plus a quarter (measured by collating it once in the
    main process).  A later batch that does not fit falls back to ordinary allocations for the tensors that overflow — slower,
    never wrong.  ``slots``: arenas in flight (default ``num_workers + 2``)."""

    def __init__(self, dataset, batch_size: int, collate_fn: Callable, num_workers: int = 4, device="cpu",
                 slots: Optional[int] = None, slot_bytes: Optional[int] = None, pin_memory: bool = True,
                 drop_last: bool = False, timeout: float = 900.0):
        self.dataset, self.batch_size, self.collate_fn = dataset, int(batch_size), collate_fn
        self.num_workers = max(0, int(num_workers))
        self.device = torch.device(device)
        self.pin_memory = bool(pin_memory) and self.device.type == "cuda"
        self.timeout = timeout
        n = len(dataset)
        self._order = [list(range(i, min(n, i + self.batch_size))) for i in range(0, n, self.batch_size)]
        if drop_last and self._order and len(self._order[-1]) < self.batch_size:
            self._order.pop()
        self._n_slots = slots if slots is not None else self.num_workers + 2
        self._slot_bytes = slot_bytes
        self._arenas: List[CollateArena] = []
        self._workers: list = []
        self._task_q = self._result_q = None
        self._side = None
        self._closed = False
        self.overflow_batches = 0

    def __len__(self) -> int:
        return len(self._order)

    @property
    def slots_pinned(self) -> int:
        """How many of the collate slots are page-locked (all of them on a GPU device unless registration failed)."""
        return sum(1 for a in self._arenas if a.pinned)

    # ---- set-up / tear-down ---------------------------------------------------------------------------------------
    def _start(self) -> None:
        if self._arenas or not self._order:
            return
        if self._slot_bytes is None:
            probe = CollateArena(0)
            with collate_into(probe):
                self.collate_fn([self.dataset[i] for i in self._order[0]])
            self._slot_bytes = int(probe.asked * 1.25) + (1 << 20)
        self._arenas = [CollateArena(self._slot_bytes, shared=self.num_workers > 0) for _ in range(max(1, self._n_slots))]
        if self.pin_memory:
            for a in self._arenas:
                a.pin()
        if self.num_workers > 0:
            ctx = torch.multiprocessing.get_context("fork")
            self._task_q, self._result_q = ctx.Queue(), ctx.Queue()
            for w in range(self.num_workers):
                p = ctx.Process(target=_worker_main, args=(w, self.dataset, self.collate_fn, self._arenas, self._task_q, self._result_q),
                                daemon=True)
                p.start()
                self._workers.append(p)
        if self.device.type == "cuda":
            self._side = torch.cuda.Stream(device=self.device)

    def close(self) -> None:
        if self._closed:
            return
        self._closed = True
        for _ in self._workers:
            try:
                self._task_q.put(None)
            except Exception:
                pass
        for p in self._workers:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        self._workers = []
        if self.device.type == "cuda":
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
        for a in self._arenas:
            a.unpin()
        self._arenas = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- iteration ------------------------------------------------------------------------------------------------------
    def __iter__(self):
        return self.batches()

    def batches(self, start: int = 0, stop: Optional[int] = None):
        """Batches ``start <= b < stop`` of the dataset order.  The workers and slots persist between calls."""
        stop = len(self._order) if stop is None else min(stop, len(self._order))
        if start >= stop:
            return
        self._start()
        free = collections.deque(range(len(self._arenas)))
        copying: collections.deque = collections.deque()       # (event, slot): H2D copies in flight
        done: Dict[int, tuple] = {}                              # batch_no -> result, finished out of order
        outstanding = 0
        state = {"dispatch": start}

        def recycle(block: bool = False):
            while copying and (copying[0][0].query() or block):
                ev, slot = copying.popleft()
                ev.synchronize()
                free.append(slot)
                block = False

        def pump():
            nonlocal outstanding
            recycle()
            while free and state["dispatch"] < stop:
                slot = free.popleft()
                b = state["dispatch"]
                state["dispatch"] += 1
                if self.num_workers > 0:
                    self._task_q.put((b, self._order[b], slot))
                    outstanding += 1
                else:
                    done[b] = self._inline(b, slot)

        def take(b: int, block: bool = True):
            """Result of batch ``b``, shipped to the device (None when it is not there yet and ``block`` is False).  Batches are
            dispatched and taken in dataset order, so a batch that is not in the workers yet is only ever waiting for a slot."""
            nonlocal outstanding
            while b not in done:
                pump()
                if b in done:
                    break
                if outstanding == 0:             # not dispatched: every slot sits behind an H2D copy (or holds a finished batch)
                    if not block:
                        return None
                    if not copying:
                        raise RuntimeError("batch loader: no free slot and nothing in flight")
                    recycle(block=True)
                    continue
                try:
                    r = self._get_result() if block else self._result_q.get_nowait()
                except queue.Empty:
                    if not block:
                        return None
                    raise
                outstanding -= 1
                done[r[0]] = r
            b_no, slot, kind, payload, err = done.pop(b)
            if err is not None:          # raised when this batch is the one being handed out, not while it is prefetched
                free.append(slot)
                return RuntimeError(f"batch loader worker failed on batch {b_no} (dataset indices {self._order[b_no][:4]}...): {err}")
            batch = _unpack(kind, payload, self._arenas[slot])
            if any(isinstance(v, torch.Tensor) and v.numel() * v.element_size() > (1 << 20) and self._arenas[slot].offset_of(v) is None
                   for v in batch.values()):
                self.overflow_batches += 1
            return self._ship(batch, slot, copying, free)

        try:
            cur = take(start)
            for b in range(start, stop):
                if isinstance(cur, Exception):
                    raise cur
                nxt = take(b + 1, block=False) if b + 1 < stop else None
                ready, ev, slot = cur
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for v in ready.values():
                        if isinstance(v, torch.Tensor) and v.is_cuda:
                            v.record_stream(torch.cuda.current_stream(self.device))
                yield ready
                if ev is None:
                    free.append(slot)            # host batch: its slot is the consumer's until the next batch is asked for
                if b + 1 < stop:
                    cur = nxt if nxt is not None else take(b + 1)
        finally:
            # an early exit leaves tasks in the workers: let them finish before the slots are used again
            while outstanding > 0 and self.num_workers > 0:
                try:
                    self._get_result()
                except Exception:
                    break
                outstanding -= 1
            if copying:
                recycle(block=True)
                while copying:
                    recycle(block=True)

    def _get_result(self):
        """Next finished batch from the workers; a worker that has died (killed, out of memory) is noticed within seconds instead of
        after the whole timeout — its batch would never arrive."""
        waited = 0.0
        while True:
            try:
                return self._result_q.get(True, 5.0)
            except queue.Empty:
                waited += 5.0
                dead = [p.pid for p in self._workers if not p.is_alive()]
                if dead:
                    raise RuntimeError(f"batch loader: worker process(es) {dead} died (exit codes "
                                       f"{[p.exitcode for p in self._workers if not p.is_alive()]}); no batch will arrive from them")
                if waited >= self.timeout:
                    raise RuntimeError(f"batch loader: no batch within {self.timeout:.0f} s")

    def _inline(self, b: int, slot: int):
        arena = self._arenas[slot]
        arena.reset()
        try:
            with collate_into(arena):
                batch = self.collate_fn([self.dataset[i] for i in self._order[b]])
            kind, payload = _pack(batch, arena)
            return (b, slot, kind, payload, None)
        except Exception as e:
            return (b, slot, None, None, f"{type(e).__name__}: {e}\n{traceback.format_exc()}")

    def _ship(self, batch, slot: int, copying, free):
        """Host batch -> (device batch, event, slot): the floating-point payload is copied on the side stream; lengths, masks and
        counts stay on the host, where the model's prompt logic reads them (a device copy would turn each read into a sync)."""
        if self.device.type != "cuda":
            return batch, None, slot
        import copy
        arena = self._arenas[slot]
        with torch.cuda.stream(self._side):
            out = copy.copy(batch)
            out.update({k: v.to(self.device, non_blocking=True) for k, v in batch.items()
                        if isinstance(v, torch.Tensor) and v.is_floating_point()})
        for k, v in batch.items():      # anything else a collate placed in the slot must not outlive the slot's recycling
            if isinstance(v, torch.Tensor) and not v.is_floating_point() and arena.offset_of(v) is not None:
                out[k] = v.clone()
        ev = torch.cuda.Event()
        ev.record(self._side)
        copying.append((ev, slot))
        return out, ev, slot

"""Zero-copy reads of list-typed audio columns of HF ``datasets`` folders.

The reference reads ``item["audio"]["array"]`` from ``dataset[idx]`` (``data/multi_task_dataset.py:135-158,229-241``).  When the
audio column is stored as plain numbers (``{"array": List(float64), "sampling_rate": int64}`` — what a folder written
without an audio codec holds, and what ``Audio``-typed columns become once cast to arrays), ``datasets`` materialises the
row by turning the Arrow ``list<double>`` into a Python list: 480 000 boxed floats for a 30 s clip, ~120 ms per row, after
which the item pipeline converts the list straight back into an array.  At the reference's 0.05 utterances/s that never
mattered; at 140 utterances/s per GPU one DataLoader worker would feed 7.

``FastAudioRows`` wraps such a dataset: ``row(idx)`` formats every OTHER column the usual way (so prompts, labels and
``few_shot_examples`` are exactly what ``dataset[idx]`` gives) and puts a ``LazyAudio`` handle where the audio cell would be;
``LazyAudio.load()`` slices the column's flat value buffer in the memory-mapped Arrow table (no copy, no Python objects) and
hands it to ``decode_audio`` like any other ``{"array", "sampling_rate"}`` cell: ~0.6 ms per 30 s clip.  Same values, same
float32 result (tests/test_host_and_abi.py pins both).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np


class LazyAudio:
    """An audio cell that has not been read yet: ``load()`` -> float32 waveform (or None for a null cell)."""

    __slots__ = ("_rows", "_column", "_index")

    def __init__(self, rows: "FastAudioRows", column: str, index: int):
        self._rows, self._column, self._index = rows, column, index

    def load(self) -> Optional[np.ndarray]:
        return self._rows.audio(self._column, self._index)


def _is_number_list_audio(arrow_type) -> bool:
    import pyarrow as pa
    if not pa.types.is_struct(arrow_type):
        return False
    names = [arrow_type.field(i).name for i in range(arrow_type.num_fields)]
    if "array" not in names:
        return False
    t = arrow_type.field(names.index("array")).type
    return (pa.types.is_list(t) or pa.types.is_large_list(t)) and (pa.types.is_floating(t.value_type) or pa.types.is_integer(t.value_type))


class FastAudioRows:
    def __init__(self, dataset, columns: List[str]):
        self.dataset = dataset
        self.columns = list(columns)
        self._rest = None           # the dataset without the audio columns (a view on the same table; built per process)
        self._chunks: Dict[str, Any] = {}

    @classmethod
    def wrap(cls, dataset) -> Optional["FastAudioRows"]:
        """A wrapper when ``dataset`` is an Arrow-backed HF dataset with at least one number-list audio column, else None
        (in-memory lists, ``Audio``-typed columns that decode from bytes and everything else keep the plain row path)."""
        table = getattr(dataset, "data", None)
        if table is None or not hasattr(dataset, "remove_columns") or not hasattr(table, "column_names"):
            return None
        try:
            cols = [n for n in table.column_names if _is_number_list_audio(table.schema.field(n).type)]
        except Exception:
            return None
        return cls(dataset, cols) if cols else None

    def __getstate__(self):           # worker processes rebuild their views (fork shares the memory map anyway)
        return {"dataset": self.dataset, "columns": self.columns, "_rest": None, "_chunks": {}}

    def __len__(self) -> int:
        return len(self.dataset)

    # ---- everything but the audio -----------------------------------------------------------------------------------
    def row(self, idx: int) -> Dict[str, Any]:
        if self._rest is None:
            self._rest = self.dataset.remove_columns(self.columns)
        idx = int(idx)
        out = dict(self._rest[idx])
        for c in self.columns:
            out[c] = LazyAudio(self, c, idx) if self._valid(c, idx) else None
        return out

    # ---- the audio --------------------------------------------------------------------------------------------------
    def _column(self, name: str):
        ent = self._chunks.get(name)
        if ent is None:
            col = self.dataset.data.column(name)
            starts = np.cumsum([0] + [len(ch) for ch in col.chunks])
            ent = self._chunks[name] = {"col": col, "starts": starts, "parts": {}}
        return ent

    def _locate(self, name: str, idx: int):
        """(chunk parts, row within the chunk) of dataset row ``idx`` — through the indices mapping of select() / shuffle()."""
        if idx < 0:
            idx += len(self.dataset)
        if not 0 <= idx < len(self.dataset):
            raise IndexError(idx)
        ind = getattr(self.dataset, "_indices", None)
        if ind is not None:
            idx = int(ind.column(0)[idx].as_py())
        ent = self._column(name)
        c = int(np.searchsorted(ent["starts"], idx, side="right") - 1)
        parts = ent["parts"].get(c)
        if parts is None:
            chunk = ent["col"].chunk(c)
            names = [chunk.type.field(i).name for i in range(chunk.type.num_fields)]
            kids = chunk.flatten()                       # children with the struct's own offset (and nulls) applied
            arr = kids[names.index("array")]
            parts = ent["parts"][c] = {
                "valid": np.asarray(chunk.is_valid()) & np.asarray(arr.is_valid()), "offsets": np.asarray(arr.offsets),
                "values": arr.values,
                "rate": kids[names.index("sampling_rate")] if "sampling_rate" in names else None}
        return parts, idx - int(ent["starts"][c])

    def _valid(self, name: str, idx: int) -> bool:
        parts, i = self._locate(name, idx)
        return bool(parts["valid"][i])

    def audio(self, name: str, idx: int) -> Optional[np.ndarray]:
        from ..utils.audio_io import decode_audio
        parts, i = self._locate(name, idx)
        if not parts["valid"][i]:
            return None
        s, e = int(parts["offsets"][i]), int(parts["offsets"][i + 1])
        values = parts["values"].slice(s, e - s).to_numpy(zero_copy_only=False)      # a view of the memory map
        rate = parts["rate"][i].as_py() if parts["rate"] is not None else None
        return decode_audio({"array": values, "sampling_rate": rate})

"""Audio cells of HF ``datasets`` rows -> mono float32 waveforms at 16 kHz.

The reference reads ``item["audio"]["array"]`` (``data/multi_task_dataset.py:135-158,401-412``), i.e. it relies on the
``datasets.Audio`` feature decoding the stored file through soundfile / torchcodec, and assumes 16 kHz.  This module accepts
every form such a cell can take on disk or in memory, so the item pipeline also works where no decoder backend is installed:

* ``{"array": ..., "sampling_rate": ...}``  — a decoded ``Audio`` cell (what the reference sees) or a plain float list;
* ``{"bytes": ..., "path": ...}``            — an ``Audio(decode=False)`` cell: RIFF/WAVE PCM (8/16/24/32-bit) or IEEE float
  (32/64-bit) is parsed here with the standard library; other containers (FLAC, MP3, OGG) need a decoder backend and raise
  a clear error;
* a bare array / list;
* a ``LazyAudio`` handle (data/arrow_audio.py): a cell of a number-list column that is read straight from the Arrow buffers.

Multi-channel audio is averaged to mono, other sampling rates are resampled to 16 kHz (polyphase, scipy) — both additive:
the reference would feed such audio unchanged.
"""
from __future__ import annotations

import io
import struct
from typing import Any, Optional

import numpy as np

TARGET_SR = 16000


def _parse_wav(buf: bytes):
    """RIFF/WAVE -> (float32 array [n, channels], sampling rate).  PCM (format 1), IEEE float (3), extensible (0xFFFE)."""
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        kind = {b"fLaC": "FLAC", b"OggS": "OGG", b"ID3": "MP3"}.get(buf[:4], None) or ("MP3" if buf[:2] in (b"\xff\xfb", b"\xff\xf3") else "unknown")
        raise ValueError(f"audio bytes are not RIFF/WAVE ({kind} container): install a datasets audio backend (soundfile / torchcodec) "
                         "or store PCM WAV")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        body = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("WAV file without fmt / data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(data, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(data[:len(data) // 2 * 2], dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(data[:len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            x = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = np.frombuffer(data[:len(data) // 4 * 4], dtype="<i4").astype(np.float32) / float(1 << 31)
        else:
            raise ValueError(f"unsupported PCM width {bits}")
    elif tag == 3:
        x = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"unsupported WAV format tag {tag}")
    n = len(x) // ch * ch
    return x[:n].reshape(-1, ch), int(sr)


def decode_audio(cell: Any, target_sr: int = TARGET_SR) -> Optional[np.ndarray]:
    """One audio cell (see module docstring) -> float32 [n] at ``target_sr``, or None for an empty cell."""
    if cell is None:
        return None
    if not isinstance(cell, (dict, np.ndarray, list, tuple)) and callable(getattr(cell, "load", None)):
        return cell.load()                           # data/arrow_audio.LazyAudio: a cell of a number-list column, read zero-copy
    sr = target_sr
    if isinstance(cell, dict):
        if cell.get("array") is not None:
            x = np.asarray(cell["array"], dtype=np.float32)
            sr = int(cell.get("sampling_rate") or target_sr)
        elif cell.get("bytes") is not None or cell.get("path"):
            raw = cell.get("bytes")
            if raw is None:
                with open(cell["path"], "rb") as f:
                    raw = f.read()
            x, sr = _parse_wav(bytes(raw))
        else:
            return None
    else:
        x = np.asarray(cell, dtype=np.float32)
    if x.ndim == 2:                                  # [n, channels] (WAV) or [channels, n] (decoded multi-channel cell)
        _note_once("downmix", "multi-channel audio cell mixed down to mono (the reference feeds cells to the model unchanged)")
        x = x.mean(axis=1 if x.shape[1] <= 8 and x.shape[0] > x.shape[1] else 0)
    x = x.reshape(-1)
    if sr != target_sr and x.size:
        _note_once("resample", f"audio cell at {sr} Hz resampled to {target_sr} Hz (the reference feeds cells to the model unchanged)")
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(target_sr))
        x = resample_poly(x.astype(np.float64), target_sr // g, sr // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


_NOTED = set()


def _note_once(key: str, msg: str) -> None:
    """One log line per kind of divergence from the reference's pass-through (not one per cell)."""
    if key not in _NOTED:
        _NOTED.add(key)
        import logging
        logging.getLogger(__name__).warning(msg)


def audio_backend_available() -> bool:
    """True when ``datasets`` can decode ``Audio`` columns itself (the reference's assumption)."""
    for mod in ("soundfile", "torchcodec"):
        try:
            __import__(mod)
            return True
        except Exception:
            continue
    return False


def undecoded_audio_columns(dataset):
    """Cast ``Audio``-typed features — top-level columns AND ``Audio`` leaves nested in dict / list / ``Sequence`` features such
    as a ``few_shot_examples`` list of dicts — to ``decode=False`` when no decoder backend is importable, so that rows carry
    ``{"bytes", "path"}`` for ``decode_audio`` instead of raising on access."""
    if audio_backend_available():
        return dataset
    try:
        from datasets import Audio, Features
    except Exception:
        return dataset

    def undecoded(feat):
        """(rewritten feature, changed?) with every decoding Audio leaf switched to decode=False."""
        if isinstance(feat, Audio):
            if getattr(feat, "decode", True):
                return Audio(sampling_rate=feat.sampling_rate, decode=False), True
            return feat, False
        if isinstance(feat, dict):
            parts = {k: undecoded(v) for k, v in feat.items()}
            return {k: v[0] for k, v in parts.items()}, any(v[1] for v in parts.values())
        if isinstance(feat, (list, tuple)) and len(feat) == 1:
            inner, ch = undecoded(feat[0])
            return [inner], ch
        inner_feat = getattr(feat, "feature", None)             # ``List`` / ``LargeList`` / legacy ``Sequence`` containers
        if inner_feat is not None:
            inner, ch = undecoded(inner_feat)
            if not ch:
                return feat, False
            import copy
            out = copy.copy(feat)                               # same container type and length, new element feature
            out.feature = inner
            return out, True
        return feat, False

    feats = getattr(dataset, "features", None) or {}
    new, changed = {}, False
    for name, feat in feats.items():
        new[name], ch = undecoded(feat)
        changed = changed or ch
    if changed:
        try:
            dataset = dataset.cast(Features(new))
        except Exception:                                     # fall back to the top-level columns only
            for name, feat in list(feats.items()):
                if isinstance(feat, Audio) and getattr(feat, "decode", True):
                    dataset = dataset.cast_column(name, Audio(sampling_rate=feat.sampling_rate, decode=False))
    return dataset

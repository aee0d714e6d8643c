"""Audio cells of HF ``datasets`` rows -> mono float32 waveforms at 16 kHz.

The reference reads ``item["audio"]["array"]`` (``data/multi_task_dataset.py:135-158,401-412``), i.e. it relies on the
``datasets.Audio`` feature decoding the stored file through soundfile / torchcodec, and assumes 16 kHz.  This module accepts
every form such a cell can take on disk or in memory, so the item pipeline also works where no decoder backend is installed:

* ``{"array": ..., "sampling_rate": ...}``  — a decoded ``Audio`` cell (what the reference sees) or a plain float list;
* ``{"bytes": ..., "path": ...}``            — an ``Audio(decode=False)`` cell: RIFF/WAVE PCM (8/16/24/32-bit) or IEEE float
  (32/64-bit) is parsed here with the standard library; other containers (FLAC, MP3, OGG) need a decoder backend and raise
  a clear error;
* a bare array / list.

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
        x = x.mean(axis=1 if x.shape[1] <= 8 and x.shape[0] > x.shape[1] else 0)
    x = x.reshape(-1)
    if sr != target_sr and x.size:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(target_sr))
        x = resample_poly(x.astype(np.float64), target_sr // g, sr // g).astype(np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


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
    """Cast ``Audio``-typed columns (top level, or inside a list of dicts such as ``few_shot_examples``) to ``decode=False`` when
    no decoder backend is importable, so that rows carry ``{"bytes", "path"}`` for ``decode_audio`` instead of raising."""
    if audio_backend_available():
        return dataset
    try:
        from datasets import Audio
    except Exception:
        return dataset
    feats = getattr(dataset, "features", None) or {}
    for name, feat in list(feats.items()):
        if isinstance(feat, Audio) and getattr(feat, "decode", True):
            dataset = dataset.cast_column(name, Audio(sampling_rate=feat.sampling_rate, decode=False))
    return dataset

"""Constant filter-bank tables handed to the front-end kernels (data, computed once on the host).

* ``slaney_mel_filters``: the Slaney-scale, area-normalised triangular bank that
  ``WhisperFeatureExtractor`` stores as ``mel_filters`` (reference use: data/model_processors.py:492-495,
  641-645) — laid out ``[n_mels, 201]`` f64 for ``icl_logmel_whisper``.
* ``kaldi_mel_banks``: the HTK-mel triangular bank of ``torchaudio.compliance.kaldi.get_mel_banks``
  (128 bins, 20 Hz .. Nyquist, 512-point FFT) with the zero Nyquist column — ``[128, 257]`` f64 for
  ``icl_fbank_kaldi`` (BEATs.preprocess; external SALMONN package).
"""
from __future__ import annotations

import math

import numpy as np


def _slaney_hz_to_mel(f: float) -> float:
    return 3.0 * f / 200.0 if f < 1000.0 else 15.0 + 27.0 * math.log(f / 1000.0) / math.log(6.4)


def _slaney_mel_to_hz(m: float) -> float:
    return 200.0 * m / 3.0 if m < 15.0 else 1000.0 * math.exp(math.log(6.4) * (m - 15.0) / 27.0)


def slaney_mel_filters(n_mels: int = 80, n_fft: int = 400, sample_rate: int = 16000) -> np.ndarray:
    n_bins = n_fft // 2 + 1
    top = _slaney_hz_to_mel(sample_rate / 2.0)
    edges = [_slaney_mel_to_hz(top * i / (n_mels + 1)) for i in range(n_mels + 2)]
    freqs = [(sample_rate // 2) * i / (n_bins - 1) for i in range(n_bins)]
    bank = np.zeros((n_mels, n_bins), dtype=np.float64)
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        norm = 2.0 / (hi - lo)
        for b, f in enumerate(freqs):
            rise, fall = (f - lo) / (mid - lo), (hi - f) / (hi - mid)
            bank[m, b] = max(0.0, min(rise, fall)) * norm
    return bank


def kaldi_mel_banks(n_mels: int = 128, n_fft: int = 512, sample_rate: int = 16000, low_freq: float = 20.0) -> np.ndarray:
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    half = n_fft // 2
    m_lo, m_hi = mel(low_freq), mel(sample_rate / 2.0)
    step = (m_hi - m_lo) / (n_mels + 1)
    bank = np.zeros((n_mels, half + 1), dtype=np.float64)
    bin_mels = [mel(sample_rate / n_fft * i) for i in range(half)]
    for m in range(n_mels):
        left, centre, right = m_lo + m * step, m_lo + (m + 1) * step, m_lo + (m + 2) * step
        for i, bm in enumerate(bin_mels):
            bank[m, i] = max(0.0, min((bm - left) / (centre - left), (right - bm) / (right - centre)))
    return bank

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the resampling step ahead of the mel stage.

PARITY UNPINNED: ``CNN implementation/preprocess.py:131-134`` delegates to
``torchaudio.transforms.Resample(orig_freq=sr, new_freq=22050)`` (torchaudio version unpinned in the
reference's README); torchaudio is not installed in the build container or on the GPU box and the
reference holds no vectors for this step.  This file restates torchaudio's published algorithm with
its defaults (``resampling_method="sinc_interp_hann"``, ``lowpass_filter_width=6``, ``rolloff=0.99``):

* rates reduced by their gcd; ``base = min(orig, new) * rolloff``; ``width = ceil(6 * orig / base)``
* filter ``p`` (one per output phase), tap ``k`` in ``[0, 2*width + orig)``:
  ``t = clamp((-p/new + (k - width)/orig) * base, -6, 6)``,
  ``h = sinc(pi t) * cos(pi t / 12)^2 * base / orig`` evaluated in float64, stored as float32
* the waveform is zero-padded by ``(width, width + orig)`` and correlated with stride ``orig``;
  output sample ``i*new + p`` is filter ``p`` applied to the window that starts at ``i*orig``
* the result is cut to ``ceil(new * length / orig)`` samples

together with the reference's own lines ``preprocess.py:127-129`` (mono mix ahead of the resampler)
and ``:151-156`` (zero pad / trim to 66 150 samples).
"""
from __future__ import annotations

import math

import numpy as np


def sinc_filters(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(filters [new, K] float32, width, orig, new) with the rates reduced by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t = np.clip(t * base, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(t == 0.0, 1.0, np.sin(t) / t)
    return (s * window * (base / orig)).astype(np.float32), width, orig, new


def resample(wave: np.ndarray, orig_freq: int, new_freq: int, dtype=np.float32) -> np.ndarray:
    """wave [..., T] -> [..., ceil(new*T/orig)]; accumulation in ``dtype`` (float64 to measure error)."""
    wave = np.asarray(wave)
    filt, width, orig, new = sinc_filters(orig_freq, new_freq)
    lead = wave.shape[:-1]
    x = wave.reshape(-1, wave.shape[-1]).astype(dtype)
    T = x.shape[1]
    K = filt.shape[1]
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    frames = (xp.shape[1] - K) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, K, axis=1)[:, ::orig][:, :frames]   # [N, frames, K]
    y = np.einsum("nfk,pk->nfp", win, filt.astype(dtype), optimize=True).reshape(x.shape[0], -1)
    target = int(math.ceil(new * T / orig))
    return y[:, :target].astype(np.float32).reshape(*lead, target)


def prepare(audio: np.ndarray, sr: int, new_freq: int = 22050, n_out: int = 66150, dtype=np.float32) -> np.ndarray:
    """audio [C, T] at ``sr`` -> [n_out]: mono mix, resample, zero pad or trim (preprocess.py:127-134,151-156)."""
    audio = np.asarray(audio, dtype=np.float32)
    mono = audio.mean(axis=0, dtype=np.float32) if audio.shape[0] > 1 else audio[0]
    y = resample(mono, sr, new_freq, dtype) if sr != new_freq else mono
    out = np.zeros(n_out, np.float32)
    n = min(n_out, y.shape[0])
    out[:n] = y[:n]
    return out

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the log-mel front end.

PARITY UNPINNED: the reference delegates this arithmetic to ``torchaudio`` (version unpinned in its
README; ``CNN implementation/preprocess.py:3-4,47-53``), which is not installed in the build
container or on the GPU box, and the reference ships no vectors for it.  This file restates
torchaudio's published semantics for ``MelSpectrogram(sample_rate=22050, n_fft=1024,
hop_length=512, n_mels=64)`` and ``AmplitudeToDB(stype="power", top_db=80)`` together with the
reference's own lines:

* ``preprocess.py:151-156``  zero-pad / trim the mono waveform to 66 150 samples
* ``preprocess.py:159``      power spectrogram (periodic Hann 1024, hop 512, centre, reflect pad,
                             one-sided) times the HTK triangular filterbank (norm=None, f 0..sr//2)
* ``preprocess.py:160``      ``10*log10(max(x,1e-10))``, floor at (max over the utterance - 80)
* ``preprocess.py:111-123``  z-score with unbiased std over the 64x130 map, ``/(std+1e-8)``, clamp +-6
* ``preprocess.py:167-168``  NaN/Inf -> 0

Two variants: ``logmel_f32`` (torch fp32, ``torch.stft``; what a CPU run of the reference would
compute, used as the timed CPU baseline) and ``logmel_f64`` (numpy float64, used to measure error).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SAMPLE_RATE = 22050
N_FFT = 1024
HOP = 512
N_MELS = 64
TARGET_LEN = SAMPLE_RATE * 3
N_FREQS = N_FFT // 2 + 1
N_FRAMES = 1 + TARGET_LEN // HOP
TOP_DB = 80.0
AMIN = 1e-10


def mel_filterbank(dtype=torch.float32) -> torch.Tensor:
    """HTK mel filterbank ``[513, 64]`` built in the order torchaudio builds it (linspace of
    frequencies, linspace in mel, triangular ``max(0, min(down, up))``, no area normalisation)."""
    all_freqs = torch.linspace(0, SAMPLE_RATE // 2, N_FREQS, dtype=dtype)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + float(SAMPLE_RATE // 2) / 700.0)
    m_pts = torch.linspace(m_min, m_max, N_MELS + 2, dtype=dtype)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


def pad_or_trim(wave: torch.Tensor) -> torch.Tensor:
    """``[..., T] -> [..., 66150]`` (preprocess.py:151-156)."""
    t = wave.shape[-1]
    if t < TARGET_LEN:
        return torch.nn.functional.pad(wave, (0, TARGET_LEN - t))
    return wave[..., :TARGET_LEN]


def logmel_f32(wave: torch.Tensor) -> torch.Tensor:
    """``[B, 66150] f32 -> [B, 1, 64, 130] f32`` with torch fp32 arithmetic."""
    wave = pad_or_trim(wave.to(torch.float32))
    win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32)
    st = torch.stft(wave, N_FFT, HOP, N_FFT, window=win, center=True, pad_mode="reflect",
                    normalized=False, onesided=True, return_complex=True)
    power = st.abs().pow(2.0)                                   # [B, 513, 130]
    mel = torch.matmul(power.transpose(-1, -2), mel_filterbank()).transpose(-1, -2)   # [B, 64, 130]
    db = 10.0 * torch.log10(torch.clamp(mel, min=AMIN))
    db = torch.maximum(db, db.amax(dim=(-2, -1), keepdim=True) - TOP_DB)
    mean = db.mean(dim=(-2, -1), keepdim=True)
    std = db.std(dim=(-2, -1), keepdim=True)                    # unbiased, as torch.std defaults
    out = torch.clamp((db - mean) / (std + 1e-8), min=-6.0, max=6.0)
    out = torch.nan_to_num(out, nan=0.0, posinf=0.0, neginf=0.0)
    return out.unsqueeze(1)


def logmel_f64(wave: np.ndarray, specmask=None) -> np.ndarray:
    """Float64 numpy evaluation of the same pipeline (fp32 filterbank values, as the reference's are
    fp32 constants).  ``[B, 66150] -> [B, 1, 64, 130] float64``.  ``specmask`` ``[B, 4]`` (first mel row,
    rows, first frame, frames) zero-fills the dB map before normalisation, where the reference applies
    its training-time masks (``preprocess.py:92-109,162-165``)."""
    x = np.asarray(wave, dtype=np.float64)
    if x.shape[-1] < TARGET_LEN:
        x = np.pad(x, ((0, 0), (0, TARGET_LEN - x.shape[-1])))
    x = x[:, :TARGET_LEN]
    pad = N_FFT // 2
    xp = np.pad(x, ((0, 0), (pad, pad)), mode="reflect")
    n = np.arange(N_FFT, dtype=np.float64)
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)
    idx = np.arange(N_FRAMES)[:, None] * HOP + np.arange(N_FFT)[None, :]
    frames = xp[:, idx] * win                                  # [B, 130, 1024]
    spec = np.fft.rfft(frames, axis=-1)
    power = spec.real ** 2 + spec.imag ** 2                    # [B, 130, 513]
    fb = mel_filterbank().numpy().astype(np.float64)           # [513, 64]
    mel = np.einsum("btf,fm->bmt", power, fb)
    db = 10.0 * np.log10(np.maximum(mel, AMIN))
    db = np.maximum(db, db.max(axis=(1, 2), keepdims=True) - TOP_DB)
    if specmask is not None:
        for b, (f0, fn, t0, tn) in enumerate(np.asarray(specmask, dtype=np.int64)):
            db[b, f0:f0 + fn, :] = 0.0
            db[b, :, t0:t0 + tn] = 0.0
    mean = db.mean(axis=(1, 2), keepdims=True)
    std = db.std(axis=(1, 2), keepdims=True, ddof=1)
    out = np.clip((db - mean) / (std + 1e-8), -6.0, 6.0)
    out = np.nan_to_num(out, nan=0.0, posinf=0.0, neginf=0.0)
    return out[:, None]

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's training-time augmentations.

Follows ``CNN implementation/preprocess.py:56-109`` (the four augmentations) and ``:137-140,162-163``
(their order inside ``AudioTransform.__call__``: time shift, noise, resample-by-index on the waveform;
frequency / time masks on the dB map before normalisation).  Every decision comes from Python's
``random`` in the reference's draw order, so seeding ``random`` reproduces a run.  The Gaussian noise
itself comes from torch's generator in the reference (``torch.randn_like``, ``:78``); here the caller
supplies the noise samples, which lets a test feed the GPU generator's own draws and compare the rest
of the arithmetic exactly.

PARITY: the waveform part is torch-only arithmetic restated from the lines above (the module that
holds it cannot be imported here: it needs torchaudio at import time), checked for self-consistency
only; there are no reference vectors for it.
"""
from __future__ import annotations

import random

import numpy as np

TARGET_LEN = 66150
N_MELS, N_FRAMES = 64, 130


class Dropped(Exception):
    """The reference's slice assignment with a zero shift raises; ``__call__`` then returns an all-zero map
    (``preprocess.py:65,70,172-174``) and makes no further draws for that utterance."""


def time_shift(x: np.ndarray, limit: float = 0.1) -> np.ndarray:
    if random.random() < 0.5:
        return x
    n = x.shape[0]
    s = int(random.random() * limit * n)
    to_right = random.random() > 0.5
    if s == 0:
        raise Dropped()
    y = np.zeros_like(x)
    if to_right:
        y[s:] = x[:-s]
    else:
        y[:-s] = x[s:]
    return y


def add_noise(x: np.ndarray, noise, factor: float = 0.005) -> np.ndarray:
    if random.random() < 0.5:
        return x
    return (x + np.asarray(noise(x.shape[0]), dtype=np.float32) * np.float32(factor)).astype(np.float32)


def pitch_shift(x: np.ndarray) -> np.ndarray:
    if random.random() < 0.5:
        return x
    f = random.uniform(0.9, 1.1)
    n = x.shape[0]
    count = int(np.ceil(n / f))
    # torch.arange(0, n, f): element i is float32(i * f) with the product taken in double
    pos = (np.arange(count, dtype=np.float64) * f).astype(np.float32)
    idx = np.clip(pos, 0, n - 1).astype(np.int64)
    return x[idx]


def draw_mask():
    """(first mel row, rows, first frame, frames); zero counts = no mask."""
    m = [0, 0, 0, 0]
    if random.random() < 0.5:
        return m
    if random.random() < 0.5:
        size = random.randint(1, 8)
        m[0], m[1] = random.randint(0, N_MELS - size), size
    if random.random() < 0.5:
        size = random.randint(1, 8)
        m[2], m[3] = random.randint(0, N_FRAMES - size), size
    return m


def augment_utterance(x: np.ndarray, noise):
    """x [n] mono at 22 050 Hz -> (wave [66150], mask [4], dropped).  ``noise(n)`` returns n N(0,1) samples."""
    try:
        y = time_shift(np.asarray(x, dtype=np.float32))
        y = add_noise(y, noise)
        y = pitch_shift(y)
    except Dropped:
        return np.zeros(TARGET_LEN, np.float32), [0, 0, 0, 0], True
    out = np.zeros(TARGET_LEN, np.float32)
    n = min(TARGET_LEN, y.shape[0])
    out[:n] = y[:n]
    return out, draw_mask(), False

"""Synthetic utterances, token ids, labels and CSV rows for tests and benchmarks.

Everything here is generated with ``numpy.random.RandomState`` so the same seeds give the same
bytes on every machine (the GPU box has neither datasets nor a tokenizer vocabulary).

Shapes follow the batch dict the reference dataset yields
(``CNN implementation/multimodal_dataset.py:229-260``):
``audio [B,1,64,130] f32``, ``input_ids [B,128] i64``, ``attention_mask [B,128] i64``,
``label [B] i64``; the waveform entering the log-mel front end is ``[B,66150] f32``
(3 s at 22 050 Hz, ``CNN implementation/preprocess.py:37-41,151-156``).
"""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 22050
N_SAMPLES = 66150          # 3 s at 22 050 Hz
SRC_RATE = 16000
SRC_SAMPLES = 64000        # 4 s at 16 kHz
N_MELS = 64
N_FRAMES = 130
MAX_LENGTH = 128
VOCAB_SIZE = 30522
CLS_ID, SEP_ID, PAD_ID = 101, 102, 0
SENTIMENTS = ("negative", "neutral", "positive")
_SEED_MOD = 2 ** 32 - 1


def _tones(idx: int, n: int, rate: int) -> np.ndarray:
    """0.1 * sum of three sinusoids + 0.01 * white noise, seeded per utterance index."""
    rs = np.random.RandomState(1234 + int(idx))
    f = rs.uniform(80.0, 4000.0, size=3)
    ph = rs.uniform(0.0, 2.0 * np.pi, size=3)
    t = np.arange(n, dtype=np.float64) / float(rate)
    x = np.zeros(n, dtype=np.float64)
    for fk, pk in zip(f, ph):
        x += np.sin(2.0 * np.pi * fk * t + pk)
    x = 0.1 * x + 0.01 * rs.standard_normal(n)
    return x.astype(np.float32)


def waveform_16k(idx: int) -> np.ndarray:
    """One 4 s / 16 kHz mono utterance, ``[64000] f32``."""
    return _tones(idx, SRC_SAMPLES, SRC_RATE)


def waveform_22k(idx: int) -> np.ndarray:
    """The same signal model sampled directly at 22 050 Hz and cut to 3 s, ``[66150] f32``.

    This is what reaches the mel stage after the reference's resample + pad/trim
    (``preprocess.py:131-156``) without needing the resampler.
    """
    return _tones(idx, N_SAMPLES, SAMPLE_RATE)


def waveform_batch(start: int, batch: int) -> np.ndarray:
    return np.stack([waveform_22k(start + i) for i in range(batch)], axis=0)


def token_batch(start: int, batch: int, max_length: int = MAX_LENGTH,
                vocab_size: int = VOCAB_SIZE, seed: int = 1234):
    """``[CLS] n~U{4..40} ids~U{1000..vocab-1} [SEP] PAD...`` -> (ids, mask), both ``[B,L] i64``."""
    ids = np.zeros((batch, max_length), dtype=np.int64)
    lo = min(1000, max(vocab_size // 2, 1))
    for i in range(batch):
        rs = np.random.RandomState((seed + 7919 * (start + i)) % _SEED_MOD)
        n = int(rs.randint(4, 41))
        n = min(n, max_length - 2)
        body = rs.randint(lo, vocab_size, size=n)
        ids[i, 0] = min(CLS_ID, vocab_size - 1)
        ids[i, 1:1 + n] = body
        ids[i, 1 + n] = min(SEP_ID, vocab_size - 1)
    mask = (ids != PAD_ID).astype(np.int64)
    return ids, mask


def label_batch(start: int, batch: int, seed: int = 1234) -> np.ndarray:
    out = np.zeros(batch, dtype=np.int64)
    for i in range(batch):
        out[i] = np.random.RandomState((seed + 104729 * (start + i) + 1) % _SEED_MOD).randint(0, 3)
    return out


def spectrogram_batch(start: int, batch: int, seed: int = 4321) -> np.ndarray:
    """Seeded stand-in log-mel features ``[B,1,64,130] f32`` (unit normal clipped to +-6)."""
    out = np.zeros((batch, 1, N_MELS, N_FRAMES), dtype=np.float32)
    for i in range(batch):
        rs = np.random.RandomState((seed + 15485863 * (start + i)) % _SEED_MOD)
        out[i, 0] = np.clip(rs.standard_normal((N_MELS, N_FRAMES)), -6.0, 6.0)
    return out


def make_batch(start: int, batch: int, vocab_size: int = VOCAB_SIZE, with_wave: bool = False):
    """Batch dict of numpy arrays with the reference's keys (plus ``wave`` when asked)."""
    ids, mask = token_batch(start, batch, vocab_size=vocab_size)
    d = {
        "audio": spectrogram_batch(start, batch),
        "input_ids": ids,
        "attention_mask": mask,
        "label": label_batch(start, batch),
    }
    if with_wave:
        d["wave"] = waveform_batch(start, batch)
    return d


def csv_rows(start: int, count: int):
    """Rows of the semicolon CSV the reference reads (``README.md:44-50``): id;sentiment;utterance."""
    labels = label_batch(start, count)
    rows = ["Audio_ID;Sentiment;Utterance"]
    for i in range(count):
        rows.append(f"utt{start + i:06d};{SENTIMENTS[int(labels[i])]};synthetic utterance {start + i}")
    return rows

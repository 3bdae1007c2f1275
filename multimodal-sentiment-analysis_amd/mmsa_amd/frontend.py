"""Log-mel front end on the GPU, behind the reference's `AudioTransform` interface.

`CNN implementation/preprocess.py:45-174` runs per utterance on CPU worker processes.  Here the mel stage
(`:151-168`: pad/trim, power spectrogram, mel filterbank, dB with top_db 80, z-score, clamp, NaN scrub) is one
batched HIP call: :class:`LogMelFrontEnd` maps ``[B, n]`` waveforms to ``[B, 1, 64, 130]``.
:class:`AudioTransform` keeps the reference's per-utterance call signature on top of it.

Input at another sample rate goes through :class:`Resampler` first (`preprocess.py:131-134`: torchaudio's sinc
resampler, one HIP launch per batch, fused with the mono mix and the pad / trim).  :class:`NoiseReducer` is the
spectral-gating noise reduction of `preprocess.py:142-148` (the default algorithm of the `noisereduce` package, which
the reference does not pin and this image does not hold: restated from the package's published description).
"""
from __future__ import annotations

import math
import os
import random

import numpy as np
import torch

from . import _lib, ops

SAMPLE_RATE = 22050
N_FFT = 1024
HOP_LENGTH = 512
N_MELS = 64
MAX_AUDIO_LENGTH = 3
TARGET_LENGTH = SAMPLE_RATE * MAX_AUDIO_LENGTH
N_FRAMES = 1 + TARGET_LENGTH // HOP_LENGTH
TOP_DB = 80.0


def set_seed(seed=42):
    """preprocess.py:26-33 (the cudnn flags have no HIP-kernel counterpart: these kernels are deterministic)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)


def htk_mel_filterbank() -> torch.Tensor:
    """[513, 64] fp32 triangular HTK filterbank, f_min 0, f_max sr//2, no normalisation -- the parameters the
    reference passes to MelSpectrogram (preprocess.py:47-52), evaluated in fp32 in the same order of operations."""
    n_freqs = N_FFT // 2 + 1
    freqs = torch.linspace(0, SAMPLE_RATE // 2, n_freqs)
    mel_lo = 2595.0 * math.log10(1.0)
    mel_hi = 2595.0 * math.log10(1.0 + float(SAMPLE_RATE // 2) / 700.0)
    pts = 700.0 * (10.0 ** (torch.linspace(mel_lo, mel_hi, N_MELS + 2) / 2595.0) - 1.0)
    width = pts[1:] - pts[:-1]
    slope = pts.unsqueeze(0) - freqs.unsqueeze(1)
    rising = (-1.0 * slope[:, :-2]) / width[:-1]
    falling = slope[:, 2:] / width[1:]
    return torch.clamp(torch.minimum(rising, falling), min=0.0).contiguous()


def _tables(device):
    import ctypes
    win = np.zeros(N_FFT, np.float32)
    t5 = np.zeros(512, np.float32)
    t10 = np.zeros(1026, np.float32)
    _lib.call("mmsa_logmel_tables_host", win.ctypes.data_as(ctypes.c_void_p), t5.ctypes.data_as(ctypes.c_void_p),
              t10.ctypes.data_as(ctypes.c_void_p))
    fb = htk_mel_filterbank()
    rng = np.zeros((N_MELS, 2), np.int32)
    nz = fb.numpy() > 0
    for m in range(N_MELS):
        idx = np.nonzero(nz[:, m])[0]
        rng[m] = (idx[0], idx[-1] + 1) if idx.size else (0, 0)
    return {
        "window": torch.from_numpy(win).to(device), "tw512": torch.from_numpy(t5).to(device),
        "tw1024": torch.from_numpy(t10).to(device), "melfb": fb.to(device),
        "melrange": torch.from_numpy(rng).to(device),
    }


class LogMelFrontEnd:
    """Batched mel stage.  ``fe(wave[B, n]) -> [B, 1, 64, 130]`` on ``device``; rows shorter than 66 150 samples are
    zero-padded, longer ones trimmed (preprocess.py:151-156)."""

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MmsaError("LogMelFrontEnd runs on an MI355X only; there is no CPU path")
        _lib.load()
        self.tables = _tables(self.device)

    def __call__(self, wave: torch.Tensor, out: torch.Tensor = None, specmask: torch.Tensor = None) -> torch.Tensor:
        if wave.dim() == 1:
            wave = wave.unsqueeze(0)
        if wave.dim() != 2:
            raise ValueError("wave must be [B, n_samples]")
        wave = wave.to(device=self.device, dtype=torch.float32)
        if wave.stride(1) != 1:
            wave = wave.contiguous()
        if out is None:
            out = torch.empty((wave.shape[0], 1, N_MELS, N_FRAMES), dtype=torch.float32, device=self.device)
        if specmask is not None:
            specmask = specmask.to(device=self.device, dtype=torch.int32).contiguous()
        return ops.logmel(wave, self.tables, out, TARGET_LENGTH, TOP_DB, specmask)


class Resampler:
    """``rs(pcm[B, C, T]) -> [B, n_out]`` at 22 050 Hz on ``device``: mono mix, band-limited sinc resampling with
    torchaudio's defaults (Hann window, lowpass_filter_width 6, rolloff 0.99) and zero pad / trim to ``n_out`` samples
    (default: the model's 66 150; ``n_out=None`` keeps the full resampled length ``ceil(22050 * T / sr)``)."""

    def __init__(self, orig_freq: int, new_freq: int = SAMPLE_RATE, device="cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MmsaError("Resampler runs on an MI355X only; there is no CPU path")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.filters = ops.resample_filters(self.orig_freq, self.new_freq, self.device)

    def resampled_length(self, T: int) -> int:
        f = self.filters
        return -(-f["new"] * int(T) // f["orig"])

    def __call__(self, pcm: torch.Tensor, n_out=TARGET_LENGTH, out: torch.Tensor = None) -> torch.Tensor:
        if pcm.dim() == 1:
            pcm = pcm[None, None]
        elif pcm.dim() == 2:
            pcm = pcm.unsqueeze(0)
        pcm = pcm.to(device=self.device, dtype=torch.float32).contiguous()
        if n_out is None:
            n_out = self.resampled_length(pcm.shape[2])
        if out is None:
            out = torch.empty((pcm.shape[0], n_out), dtype=torch.float32, device=self.device)
        return ops.resample(pcm, self.filters, out)


class NoiseReducer:
    """``nr(wave[B, T], lens=None) -> [B, T]`` on ``device``: the non-stationary spectral gate that
    ``noisereduce.reduce_noise(y, sr)`` applies with its defaults (preprocess.py:142-148) -- STFT (n_fft 1024, hop 256),
    magnitude smoothed along time with a 2 s time constant, sigmoid gate two smoothed magnitudes above it, mask smoothed
    over 500 Hz x 50 ms, inverse STFT -- for a whole batch in four launches.  ``lens`` [B]: valid lengths of rows
    holding utterances of different lengths; each is gated as a signal of its own length."""

    def __init__(self, sr: int = SAMPLE_RATE, device="cuda", tables=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.MmsaError("NoiseReducer runs on an MI355X only; there is no CPU path")
        _lib.load()
        self.tables = tables if tables is not None else _tables(self.device)
        self.filters = ops.denoise_filters(int(sr), self.device)

    def __call__(self, wave: torch.Tensor, lens=None, out: torch.Tensor = None) -> torch.Tensor:
        if wave.dim() == 1:
            wave = wave.unsqueeze(0)
        if wave.dim() != 2:
            raise ValueError("wave must be [B, n_samples]")
        wave = wave.to(device=self.device, dtype=torch.float32)
        if wave.stride(1) != 1:
            wave = wave.contiguous()
        if lens is not None:
            host = torch.as_tensor(lens, dtype=torch.int64).cpu()
            # checked on the host: the kernels index rows by these lengths
            if host.shape != (wave.shape[0],) or int(host.min()) < 1 or int(host.max()) > wave.shape[1]:
                raise ValueError("lens must hold one length in 1..n_samples per row")
            lens = host.to(dtype=torch.int32, device=self.device).contiguous()
        return ops.denoise(wave, self.filters, self.tables, out=out, lens=lens)


class AudioTransform:
    """Per-utterance interface of the reference (``transform(audio[C, T], sr) -> [1, 64, 130]`` CPU tensor).

    The spectrogram is computed by the HIP front end, so call it from the process that owns the GPU
    (``DataLoader(num_workers=0)``), or use ``waveform_only=True`` in worker processes -- they then return the
    prepared ``[1, 66150]`` waveform and the training process runs :class:`LogMelFrontEnd` on whole batches.
    Waveform augmentations follow preprocess.py:56-90 with the same ``random`` draw order.  ``noise_reduce`` (on, as
    in the reference, `:142-148`) runs :class:`NoiseReducer` between the augmentations and the pad / trim; it needs the
    GPU, so worker processes that must not touch it pass ``noise_reduce=False`` and leave that step to the training
    process.
    """

    def __init__(self, augment=False, device="cuda", waveform_only=False, strict=False, noise_reduce=True):
        self.augment = augment
        self.device = device
        self.waveform_only = waveform_only
        self.strict = strict          # True: errors propagate; False: the reference's contract (log, return zeros)
        self.noise_reduce = noise_reduce
        self._fe = None
        self._nr = None
        self._rs = {}

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_fe"] = None            # device tables are rebuilt lazily in the receiving process
        d["_nr"] = None
        d["_rs"] = {}
        return d

    def _front_end(self):
        if self._fe is None:
            self._fe = LogMelFrontEnd(self.device)
        return self._fe

    def _reducer(self):
        if self._nr is None:
            fe = self._front_end()
            self._nr = NoiseReducer(SAMPLE_RATE, fe.device, fe.tables)
        return self._nr

    # -- augmentations (host side, training only) ------------------------------------------------
    def time_shift(self, audio, shift_limit=0.1):
        if random.random() < 0.5:
            return audio
        shift = int(random.random() * shift_limit * audio.shape[1])
        right = random.random() > 0.5
        out = torch.zeros_like(audio)
        n = audio.shape[1]
        if shift == 0:
            # the reference's slice assignment (`audio_shift[:, 0:] = audio[:, :-0]`, preprocess.py:65,70) has an empty
            # right-hand side for a zero shift and raises; its __call__ answers with an all-zero map and draws no more
            raise RuntimeError("time shift of zero samples: the reference's slice assignment fails for it")
        if right:
            out[:, shift:] = audio[:, :n - shift]
        else:
            out[:, :n - shift] = audio[:, shift:]
        return out

    def add_noise(self, audio, noise_factor=0.005):
        if random.random() < 0.5:
            return audio
        return audio + torch.randn_like(audio) * noise_factor

    def pitch_shift(self, audio, sr):
        if random.random() < 0.5:
            return audio
        factor = random.uniform(0.9, 1.1)
        idx = torch.arange(0, audio.shape[1], factor).clamp(0, audio.shape[1] - 1).long()
        return audio[:, idx]

    def spectral_mask(self, spec, max_mask_freq=8, max_mask_time=8):
        if random.random() < 0.5:
            return spec
        if random.random() < 0.5:
            size = random.randint(1, max_mask_freq)
            start = random.randint(0, spec.shape[1] - size)
            spec[:, start:start + size, :] = 0
        if random.random() < 0.5:
            size = random.randint(1, max_mask_time)
            start = random.randint(0, spec.shape[2] - size)
            spec[:, :, start:start + size] = 0
        return spec

    # -- augmentations (whole batches on the GPU) -------------------------------------------------
    @staticmethod
    def draw_augmentation(n_in: int):
        """One utterance's random decisions in the reference's draw order (preprocess.py:56-109,137-140,162-163):
        (wave row [shift mode, shift, noise on, index step], mask row [first mel row, rows, first frame, frames]).
        A zero shift makes the reference's slice assignment raise, which its ``__call__`` answers with an all-zero
        map and no further draws (`:65,70,172-174`): encoded as mode != 0 with shift 0, everything else off."""
        row, mask = [0.0, 0.0, 0.0, 0.0], [0, 0, 0, 0]
        if random.random() >= 0.5:
            shift = int(random.random() * 0.1 * n_in)
            row[0] = 1.0 if random.random() > 0.5 else 2.0
            row[1] = float(shift)
            if shift == 0:
                return row, mask
        if random.random() >= 0.5:
            row[2] = 1.0
        if random.random() >= 0.5:
            row[3] = random.uniform(0.9, 1.1)
        if random.random() >= 0.5:
            if random.random() < 0.5:
                size = random.randint(1, 8)
                mask[0], mask[1] = random.randint(0, N_MELS - size), size
            if random.random() < 0.5:
                size = random.randint(1, 8)
                mask[2], mask[3] = random.randint(0, N_FRAMES - size), size
        return row, mask

    def augment_batch(self, waves: torch.Tensor, rows=None, masks=None, seed=None, return_wave=False):
        """``waves [B, n]`` (mono, 22 050 Hz, un-padded) -> ``[B, 1, 64, 130]`` on the GPU: the reference's
        augmented training transform for a whole batch in two launches (waveform augmentation + pad/trim, then the
        mel stage with the spectral masks).  ``rows`` / ``masks`` default to fresh draws, utterance by utterance."""
        fe = self._front_end()
        dev = fe.device
        waves = waves.to(device=dev, dtype=torch.float32)
        if waves.stride(1) != 1:
            waves = waves.contiguous()
        B, n = waves.shape
        if rows is None:
            drawn = [self.draw_augmentation(n) for _ in range(B)]
            rows, masks = [d[0] for d in drawn], [d[1] for d in drawn]
        if seed is None:
            # the reference consumes torch's generator for the noise (randn_like); so does the seed of ours
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        params = torch.tensor(rows, dtype=torch.float64).reshape(B, 4).to(dev)
        if self.noise_reduce:
            # the gate sees each utterance at its augmented length, before the pad / trim (preprocess.py:137-156)
            lens = [n if r[3] <= 0 else int(math.ceil(n / r[3])) for r in rows]
            aug = torch.empty((B, max(lens)), dtype=torch.float32, device=dev)
            ops.augment_wave(waves, params, aug, 0.005, seed)
            aug = self._reducer()(aug, lens=lens)
            out = torch.zeros((B, TARGET_LENGTH), dtype=torch.float32, device=dev)
            m = min(aug.shape[1], TARGET_LENGTH)
            out[:, :m] = aug[:, :m]
        else:
            out = torch.empty((B, TARGET_LENGTH), dtype=torch.float32, device=dev)
            ops.augment_wave(waves, params, out, 0.005, seed)
        if return_wave:
            return out
        mk = None if masks is None else torch.tensor(masks, dtype=torch.int32).reshape(B, 4)
        return fe(out, specmask=mk)

    def prepare_waveform(self, audio: torch.Tensor, sr: int = SAMPLE_RATE) -> torch.Tensor:
        """Everything before the mel stage: mono mix, resampling, augmentation, noise reduction, pad/trim."""
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        if sr != SAMPLE_RATE:
            # preprocess.py:127-134: mono mix and resampling in one HIP launch (needs the process that owns the GPU)
            if sr not in self._rs:
                self._rs[sr] = Resampler(sr, SAMPLE_RATE, self.device)
            # augmentations and the noise gate see the full resampled length
            n_out = None if (self.augment or self.noise_reduce) else TARGET_LENGTH
            audio = self._rs[sr](audio, n_out=n_out).cpu()
        elif audio.shape[0] > 1:
            audio = audio.mean(dim=0, keepdim=True)
        if self.augment:
            audio = self.time_shift(audio)
            audio = self.add_noise(audio)
            audio = self.pitch_shift(audio, SAMPLE_RATE)
        if self.noise_reduce and audio.shape[1] > 0:
            audio = self._reducer()(audio).cpu()
        n = audio.shape[1]
        if n < TARGET_LENGTH:
            audio = torch.nn.functional.pad(audio, (0, TARGET_LENGTH - n))
        elif n > TARGET_LENGTH:
            audio = audio[:, :TARGET_LENGTH]
        return audio

    def __call__(self, audio, sr=SAMPLE_RATE):
        """``transform(audio[C, T], sr) -> [1, 64, 130]``.  Like the reference (preprocess.py:172-174) a failing
        utterance is logged and answered with an all-zero map -- except for a missing / broken HIP library
        (:class:`MmsaError`), which always propagates, and unless ``strict=True``."""
        try:
            return self._transform(audio, sr)
        except _lib.MmsaError:
            raise
        except Exception as e:
            if self.strict:
                raise
            import logging
            logging.getLogger(__name__).error(f"Error in audio transformation: {e}")
            if self.waveform_only:
                return torch.zeros((1, TARGET_LENGTH))
            return torch.zeros((1, N_MELS, N_FRAMES))

    def _transform(self, audio, sr):
        if self.waveform_only:
            return self.prepare_waveform(audio, sr)
        # mono mix / resampling, every augmentation (waveform ones in one launch, the spectral masks inside the mel
        # stage ahead of the normalisation: preprocess.py:137-140,162-165), the noise gate and the mel stage on the GPU
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        if sr != SAMPLE_RATE:
            if sr not in self._rs:
                self._rs[sr] = Resampler(sr, SAMPLE_RATE, self.device)
            full = self.augment or self.noise_reduce       # both see the whole resampled utterance
            audio = self._rs[sr](audio, n_out=None if full else TARGET_LENGTH)
        elif audio.shape[0] > 1:
            audio = audio.mean(dim=0, keepdim=True)
        if self.augment:
            return self.augment_batch(audio)[0].cpu()
        if self.noise_reduce and audio.shape[1] > 0:
            audio = self._reducer()(audio)
        return self._front_end()(audio)[0].cpu()


def calculate_class_weights(csv_path):
    """'balanced' class weights n / (k * count_c) over the CSV's Sentiment column (preprocess.py:259-292);
    falls back to ones if the file cannot be read, as the reference does."""
    label_map = {"negative": 0, "-1": 0, "-1.0": 0, "neutral": 1, "0": 1, "0.0": 1, "positive": 2, "1": 2, "1.0": 2}
    try:
        with open(csv_path) as f:
            header = f.readline().rstrip("\n").split(";")
            col = header.index("Sentiment")
            y = [label_map.get(line.rstrip("\n").split(";")[col].strip(), 1) for line in f if line.strip()]
        classes, counts = np.unique(np.asarray(y), return_counts=True)
        w = len(y) / (len(classes) * counts.astype(np.float64))
        return torch.tensor(w, dtype=torch.float)
    except Exception:
        return torch.tensor([1.0, 1.0, 1.0], dtype=torch.float)

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
    (``DataLoader(num_workers=0)``).  Inside a DataLoader worker process a call that would need the GPU raises
    :class:`MmsaError` (never the all-zero map): the reference's loader runs its transform in four worker processes
    (multimodal_train.py:435-442), which must not touch the GPU.  For them :meth:`for_workers` gives the CPU-only half
    (decoded PCM -> mono utterance, picklable, fork-safe) and :meth:`process_batch` / :class:`BatchFrontEnd` the GPU
    half, run on each collated batch in the training process: resampling, augmentations, noise gate, pad / trim and
    the mel stage for the whole batch.
    Waveform augmentations follow preprocess.py:56-90 with the same ``random`` draw order.  ``noise_reduce`` (on, as
    in the reference, `:142-148`) runs :class:`NoiseReducer` between the augmentations and the pad / trim.
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

    def for_workers(self, max_samples=None) -> "WorkerAudioTransform":
        """The transform to hand to the dataset when the loader uses worker processes; see
        :class:`WorkerAudioTransform`."""
        return WorkerAudioTransform(max_samples)

    def batch_front_end(self) -> "BatchFrontEnd":
        return BatchFrontEnd(self)

    def _needs_gpu(self, sr) -> bool:
        return (not self.waveform_only) or sr != SAMPLE_RATE or self.noise_reduce

    def _front_end(self):
        if self._fe is None:
            self._fe = LogMelFrontEnd(self.device)
        return self._fe

    def _resampler(self, sr):
        if sr not in self._rs:
            self._rs[sr] = Resampler(sr, SAMPLE_RATE, self.device)
        return self._rs[sr]

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
        return self.process_batch(waves, None, SAMPLE_RATE, rows, masks, seed, return_wave, augment=True)

    def process_batch(self, pcm: torch.Tensor, lengths=None, sr: int = SAMPLE_RATE, rows=None, masks=None, seed=None,
                      return_wave=False, augment=None):
        """The reference's whole per-utterance transform (preprocess.py:125-168) for a batch, on the GPU.

        ``pcm [B, T]``: mono utterances at ``sr`` Hz, rows zero-padded to a common ``T``; ``lengths [B]``: samples each
        row really holds (default: all ``T``).  Resampling to 22 050 Hz, then -- each utterance at its own length, as
        the reference sees it -- the waveform augmentations (``augment``; ``rows`` / ``masks`` default to fresh draws
        in the reference's order, utterance by utterance) and the noise gate, then pad / trim to 66 150 samples and
        the mel stage.  Returns ``[B, 1, 64, 130]`` on the device (``return_wave``: the ``[B, 66150]`` waveforms)."""
        augment = self.augment if augment is None else augment
        fe = self._front_end()
        dev = fe.device
        if pcm.dim() != 2:
            raise ValueError("pcm must be [B, T] (mono rows)")
        x = pcm.to(device=dev, dtype=torch.float32)
        if x.stride(1) != 1:
            x = x.contiguous()
        B, n = x.shape
        if lengths is None:
            lens = [n] * B
        else:
            lens = [int(v) for v in torch.as_tensor(lengths).flatten().tolist()]
            if len(lens) != B or min(lens) < 0 or max(lens) > n:
                raise ValueError("lengths must hold one count in 0..T per row")
            lens = [max(v, 1) for v in lens]      # an empty utterance is one sample of silence (rows are zero-padded)
        if sr != SAMPLE_RATE:
            rs = self._resampler(sr)
            x = rs(x.unsqueeze(1), n_out=None)
            lens = [rs.resampled_length(v) for v in lens]
            n = x.shape[1]
        ragged = any(v != n for v in lens)

        def on_device(vals):
            return torch.tensor(vals, dtype=torch.int32).to(dev)

        if ragged and sr != SAMPLE_RATE:
            # what the filter rings out behind a shorter row's end belongs to no utterance
            x.masked_fill_(torch.arange(n, device=dev)[None, :] >= on_device(lens)[:, None], 0.0)
        if augment:
            if rows is None:
                drawn = [self.draw_augmentation(v) for v in lens]
                rows, masks = [d[0] for d in drawn], [d[1] for d in drawn]
            if seed is None:
                # the reference consumes torch's generator for the noise (randn_like); so does the seed of ours
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            params = torch.tensor(rows, dtype=torch.float64).reshape(B, 4).to(dev)
            lens_in = on_device(lens) if ragged else None
            if self.noise_reduce:
                # the gate sees each utterance at its augmented length, before the pad / trim (preprocess.py:137-156)
                lens = [v if r[3] <= 0 else int(math.ceil(v / r[3])) for v, r in zip(lens, rows)]
                aug = torch.empty((B, max(lens)), dtype=torch.float32, device=dev)
                ops.augment_wave(x, params, aug, 0.005, seed, lens_in)
                out = self._gate_pad_trim(aug, lens)
            else:
                out = torch.empty((B, TARGET_LENGTH), dtype=torch.float32, device=dev)
                ops.augment_wave(x, params, out, 0.005, seed, lens_in)
        elif self.noise_reduce:
            out = self._gate_pad_trim(x, lens if ragged else None)
        else:
            out = x                  # the mel stage pads / trims
        if return_wave:
            if out.shape[1] != TARGET_LENGTH:
                full = torch.zeros((B, TARGET_LENGTH), dtype=torch.float32, device=dev)
                m = min(out.shape[1], TARGET_LENGTH)
                full[:, :m] = out[:, :m]
                out = full
            return out
        mk = None if (masks is None or not augment) else torch.tensor(masks, dtype=torch.int32).reshape(B, 4)
        return fe(out, specmask=mk)

    def _gate_pad_trim(self, x, lens):
        """Noise gate over rows of (optionally) different lengths, then zero pad / trim to 66 150 samples."""
        B, n = x.shape
        nr = self._reducer()
        if lens is not None and n > 600000:
            # rows of different lengths are gated in one chunk; longer batches go utterance by utterance
            y = torch.zeros_like(x)
            for b, v in enumerate(lens):
                y[b:b + 1, :v] = nr(x[b:b + 1, :v])
        else:
            y = nr(x, lens=lens)
        out = torch.zeros((B, TARGET_LENGTH), dtype=torch.float32, device=x.device)
        m = min(n, TARGET_LENGTH)
        out[:, :m] = y[:, :m]
        return out

    def prepare_waveform(self, audio: torch.Tensor, sr: int = SAMPLE_RATE) -> torch.Tensor:
        """Everything before the mel stage: mono mix, resampling, augmentation, noise reduction, pad/trim."""
        if audio.dim() == 1:
            audio = audio.unsqueeze(0)
        if sr != SAMPLE_RATE:
            # preprocess.py:127-134: mono mix and resampling in one HIP launch (needs the process that owns the GPU)
            # augmentations and the noise gate see the full resampled length
            n_out = None if (self.augment or self.noise_reduce) else TARGET_LENGTH
            audio = self._resampler(sr)(audio, n_out=n_out).cpu()
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
        utterance is logged and answered with an all-zero map, unless ``strict=True`` -- and except where the failure
        is not the utterance's: a missing / broken HIP library, a GPU that cannot be used from this process, or a call
        from a DataLoader worker process that would need the GPU.  Those raise :class:`MmsaError` whatever ``strict``
        says, because an all-zero map for every utterance would let a training run "succeed" on silence."""
        if self._needs_gpu(sr):
            from torch.utils.data import get_worker_info
            if get_worker_info() is not None:
                raise _lib.MmsaError(
                    "AudioTransform was called inside a DataLoader worker process and needs the GPU there (the mel "
                    "stage, the resampler and the noise gate are HIP kernels). Give the dataset "
                    "AudioTransform(...).for_workers() instead and run AudioTransform(...).batch_front_end() on each "
                    "collated batch in the training process (collate_fn=mmsa_amd.frontend.collate_utterances), or "
                    "use DataLoader(num_workers=0).")
        try:
            return self._transform(audio, sr)
        except _lib.MmsaError:
            raise
        except Exception as e:
            if _gpu_unusable(e):
                raise _lib.MmsaError(f"the GPU cannot be used from this process: {e}") from e
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
            full = self.augment or self.noise_reduce       # both see the whole resampled utterance
            audio = self._resampler(sr)(audio, n_out=None if full else TARGET_LENGTH)
        elif audio.shape[0] > 1:
            audio = audio.mean(dim=0, keepdim=True)
        if self.augment:
            return self.augment_batch(audio)[0].cpu()
        if self.noise_reduce and audio.shape[1] > 0:
            audio = self._reducer()(audio)
        return self._front_end()(audio)[0].cpu()


def _gpu_unusable(e: Exception) -> bool:
    """True for the errors torch raises when the GPU runtime cannot be brought up in this process (no device, no
    driver, a forked child of a process that already initialised it): not a property of one utterance."""
    if not isinstance(e, (RuntimeError, AssertionError)):
        return False
    msg = str(e)
    return any(k in msg for k in ("CUDA", "HIP", "cuda runtime", "hipError", "NVIDIA driver", "forked subprocess"))


class WorkerAudioTransform:
    """CPU-only half of :class:`AudioTransform` for ``DataLoader`` worker processes (multimodal_train.py:435-442 runs the
    dataset's transform in four of them, multimodal_dataset.py:172): holds no device object, is picklable and
    fork-safe, and never touches the GPU.

    ``t(audio[C, T], sr) -> {"pcm": [T'] float32, "length": int64 scalar, "sr": int64 scalar}``: the mono mix of
    preprocess.py:127-129, nothing else.  With ``max_samples`` every utterance is zero-padded / cut to that many
    samples (at the file's rate), so the loader's default collate can stack them; without it utterances keep their
    length and the loader needs ``collate_fn=collate_utterances``, which pads each batch to its longest row.
    Everything after the mono mix -- resampling, augmentations, noise gate, pad / trim, mel stage -- runs on the GPU
    on the collated batch: :class:`BatchFrontEnd`.  Like the reference's transform (`:172-174`) it never raises: an
    utterance it cannot read becomes three seconds of silence, whose map is all zeros."""

    def __init__(self, max_samples=None):
        if max_samples is not None and int(max_samples) < 1:
            raise ValueError("max_samples must be positive")
        self.max_samples = None if max_samples is None else int(max_samples)

    def __call__(self, audio, sr=SAMPLE_RATE):
        try:
            audio = torch.as_tensor(audio)
            if audio.dim() == 1:
                audio = audio.unsqueeze(0)
            if audio.dim() != 2:
                raise ValueError("audio must be [channels, samples]")
            pcm = audio.to(torch.float32)
            pcm = pcm.mean(dim=0) if pcm.shape[0] > 1 else pcm[0]
            sr = int(sr)
            if sr <= 0:
                raise ValueError("sample rate must be positive")
        except Exception as e:
            import logging
            logging.getLogger(__name__).error(f"Error in audio transformation: {e}")
            pcm, sr = torch.zeros(TARGET_LENGTH), SAMPLE_RATE
        n = pcm.shape[0]
        if self.max_samples is not None:
            if n >= self.max_samples:
                pcm, n = pcm[:self.max_samples], self.max_samples
            else:
                pcm = torch.nn.functional.pad(pcm, (0, self.max_samples - n))
        return {"pcm": pcm.contiguous(), "length": torch.tensor(n, dtype=torch.int64),
                "sr": torch.tensor(sr, dtype=torch.int64)}


def _pad_utterances(items):
    """List of per-utterance dicts (or the dataset's fallback tensors) -> one batch dict with rows padded to the
    longest utterance."""
    rows, lens, rates = [], [], []
    for it in items:
        if isinstance(it, dict):
            n = int(it["length"])
            rows.append(it["pcm"][:n] if it["pcm"].shape[0] != n else it["pcm"])
            lens.append(n)
            rates.append(int(it["sr"]))
        else:
            # multimodal_dataset.py:175-178 answers its own failures with a zero map: here, silence
            rows.append(torch.zeros(TARGET_LENGTH))
            lens.append(TARGET_LENGTH)
            rates.append(SAMPLE_RATE)
    T = max(max(lens), 1)
    pcm = torch.zeros((len(rows), T), dtype=torch.float32)
    for i, r in enumerate(rows):
        pcm[i, :r.shape[0]] = r
    return {"pcm": pcm, "length": torch.tensor(lens, dtype=torch.int64), "sr": torch.tensor(rates, dtype=torch.int64)}


def collate_utterances(samples):
    """``collate_fn`` for a loader whose dataset uses :class:`WorkerAudioTransform` without ``max_samples``: the
    ``'audio'`` entries (utterances of different lengths) are padded to the longest of the batch, everything else goes
    through torch's default collate."""
    from torch.utils.data import default_collate
    first = samples[0]
    if isinstance(first, dict) and "pcm" in first:
        return _pad_utterances(samples)
    if isinstance(first, dict) and "audio" in first:
        out = default_collate([{k: v for k, v in s.items() if k != "audio"} for s in samples])
        out["audio"] = _pad_utterances([s["audio"] for s in samples])
        return out
    return default_collate(samples)


class BatchFrontEnd:
    """GPU half of the transform, called in the training process on what the loader collated from
    :class:`WorkerAudioTransform` outputs: ``bfe(batch["audio"]) -> [B, 1, 64, 130]`` on the device, where the
    reference's loop has ``batch['audio'].to(DEVICE)`` (multimodal_train.py:160,165).  Utterances are grouped by sample
    rate (one group in practice) and each group runs :meth:`AudioTransform.process_batch`; augmentation decisions are
    drawn here, utterance by utterance in batch order, from Python's ``random`` as the reference draws them."""

    def __init__(self, transform: "AudioTransform" = None, **kw):
        self.transform = transform if transform is not None else AudioTransform(**kw)

    def __call__(self, audio, return_wave=False):
        if torch.is_tensor(audio):
            if audio.dim() == 4:
                return audio.to(self.transform.device)        # already spectrograms
            audio = {"pcm": audio, "length": torch.full((audio.shape[0],), audio.shape[1]),
                     "sr": torch.full((audio.shape[0],), SAMPLE_RATE)}
        pcm, length, sr = audio["pcm"], audio["length"].flatten().tolist(), audio["sr"].flatten().tolist()
        if pcm.dim() != 2 or len(length) != pcm.shape[0] or len(sr) != pcm.shape[0]:
            raise ValueError("expected {'pcm': [B, T], 'length': [B], 'sr': [B]}")
        t = self.transform
        rates = sorted(set(int(v) for v in sr))
        if len(rates) == 1:
            return t.process_batch(pcm, length, rates[0], return_wave=return_wave)
        shape = (pcm.shape[0], TARGET_LENGTH) if return_wave else (pcm.shape[0], 1, N_MELS, N_FRAMES)
        out = torch.empty(shape, dtype=torch.float32, device=t._front_end().device)
        # draws stay in batch order: decisions first, for all utterances, then one launch sequence per rate
        rs_len = [v if int(r) == SAMPLE_RATE else t._resampler(int(r)).resampled_length(max(v, 1))
                  for v, r in zip(length, sr)]
        drawn = [t.draw_augmentation(max(v, 1)) for v in rs_len] if t.augment else None
        for rate in rates:
            idx = [i for i, r in enumerate(sr) if int(r) == rate]
            sel = torch.tensor(idx)
            n = max(max(length[i] for i in idx), 1)
            rows = masks = None
            if drawn is not None:
                rows, masks = [drawn[i][0] for i in idx], [drawn[i][1] for i in idx]
            res = t.process_batch(pcm[sel, :n], [length[i] for i in idx], rate, rows, masks, return_wave=return_wave)
            out[sel.to(out.device)] = res
        return out


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

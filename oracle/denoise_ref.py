"""CPU restatement of the noise reduction of `CNN implementation/preprocess.py:142-148`. Test infrastructure only:
nothing on the product path may import this file.

The reference calls ``noisereduce.reduce_noise(y=audio, sr=22050)`` with every other argument at its default.  The
`noisereduce` package is a third-party dependency that is not vendored in the reference, not pinned by it
(`README.md:20` installs it unversioned) and absent from this image, and the reference holds no test or fixture at
this boundary: PARITY UNPINNED.  What is restated here is the package's published default algorithm (releases 2.0.1 to
3.0.x, keyword interface ``y=, sr=``): non-stationary spectral gating,

  1. the signal is placed in a zero buffer with 30 000 samples of padding on each side (one chunk up to 600 000
     samples; longer signals chunk by chunk, the padding then showing the neighbouring samples), in float64;
  2. ``scipy.signal.stft`` with nperseg = nfft = 1024, hop 256, periodic Hann window, zero boundary extension, no
     end padding, 'spectrum' scaling;
  3. |X| is smoothed along time by a forward-backward one-pole filter (``filtfilt([b], [1, b-1], padtype=None)``)
     whose coefficient follows from the time constant of 2 s: t = 2 * sr / hop frames,
     b = (sqrt(1 + 4 t^2) - 1) / (2 t^2);
  4. mask = sigmoid(((|X| - smooth) / smooth - 2) * 10);
  5. the mask is smoothed by ``fftconvolve(mask, filt, 'same')`` with the normalised outer product of two triangular
     ramps: int(500 Hz / (sr / 512)) = 11 steps along frequency (23 taps), int(50 ms / (1000 * hop / sr)) = 4 steps
     along time (9 taps);
  6. mask * prop_decrease + (1 - prop_decrease) with prop_decrease 1, X * mask, ``scipy.signal.istft``, padding
     removed, result cast to the input dtype.

The STFT, inverse STFT, filtfilt and convolution are scipy's own routines, as in the package; only the wiring above is
restated.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import fftconvolve, filtfilt, istft, stft

N_FFT = 1024
HOP = 256
PADDING = 30000
CHUNK = 600000


def smoothing_taps(sr: int, n_fft: int = N_FFT, hop: int = HOP, freq_mask_smooth_hz: float = 500.0,
                   time_mask_smooth_ms: float = 50.0):
    """The two triangular ramps of the mask-smoothing filter (un-normalised): frequency taps, time taps."""
    n_grad_freq = int(freq_mask_smooth_hz / (sr / (n_fft / 2)))
    n_grad_time = int(time_mask_smooth_ms / ((hop / sr) * 1000))
    if n_grad_freq < 1 or n_grad_time < 1:
        raise ValueError("smoothing width below one bin / one frame")

    def ramp(n):
        return np.concatenate([np.linspace(0, 1, n + 1, endpoint=False), np.linspace(1, 0, n + 2)])[1:-1]

    return ramp(n_grad_freq), ramp(n_grad_time)


def iir_coefficient(sr: int, hop: int = HOP, time_constant_s: float = 2.0) -> float:
    t_frames = time_constant_s * sr / float(hop)
    return (np.sqrt(1 + 4 * t_frames ** 2) - 1) / (2 * t_frames ** 2)


def _gate_chunk(chunk: np.ndarray, sr: int, thresh: float, slope: float, prop_decrease: float) -> np.ndarray:
    """One zero-padded float64 chunk [n] -> the gated chunk [n] (trailing samples past the last whole hop stay 0)."""
    _, _, spec = stft(chunk, nfft=N_FFT, noverlap=N_FFT - HOP, nperseg=N_FFT, padded=False)
    mag = np.abs(spec)
    b = iir_coefficient(sr)
    smooth = filtfilt([b], [1, b - 1], mag, axis=-1, padtype=None)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        above = (mag - smooth) / smooth
        mask = 1 / (1 + np.exp(-(above - thresh) * slope))
    vf, vt = smoothing_taps(sr)
    filt = np.outer(vf, vt)
    filt = filt / np.sum(filt)
    mask = fftconvolve(mask, filt, mode="same")
    mask = mask * prop_decrease + np.ones(mask.shape) * (1.0 - prop_decrease)
    _, rec = istft(spec * mask, nfft=N_FFT, noverlap=N_FFT - HOP, nperseg=N_FFT)
    out = np.zeros(chunk.shape, chunk.dtype)
    out[:len(rec)] = rec[:len(out)]
    return out


def reduce_noise(y: np.ndarray, sr: int = 22050, thresh: float = 2.0, slope: float = 10.0,
                 prop_decrease: float = 1.0, chunk_size: int = CHUNK, padding: int = PADDING) -> np.ndarray:
    """[n] mono signal -> [n] denoised signal in y's dtype."""
    y = np.asarray(y)
    n = y.shape[0]
    out = np.zeros(n, np.float64)
    for c0 in range(0, n, chunk_size):
        L = min(chunk_size, n - c0)
        chunk = np.zeros(L + 2 * padding)           # float64, as the package's chunk buffer
        lo, hi = max(0, c0 - padding), min(n, c0 + L + padding)
        chunk[lo - (c0 - padding):hi - (c0 - padding)] = y[lo:hi]
        out[c0:c0 + L] = _gate_chunk(chunk, sr, thresh, slope, prop_decrease)[padding:padding + L]
    return out.astype(y.dtype)

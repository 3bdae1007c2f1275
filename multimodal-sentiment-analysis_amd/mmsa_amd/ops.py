"""Thin tensor-level wrappers over the C ABI (one Python function per entry point family).

Every wrapper checks device / dtype / contiguity and the shapes the kernels assume before launching, then
enqueues on torch's current HIP stream.  No wrapper computes anything itself and none has a fallback.
"""
from __future__ import annotations

import ctypes
import math
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib

F32 = torch.float32
EPI_NONE, EPI_LN_RELU, EPI_RELU_LN, EPI_RELU = 0, 1, 2, 3
EPI_LN, EPI_TANH, EPI_SIGMOID, EPI_LN_GELU = 4, 5, 6, 7
BN_MOMENTUM, BN_EPS = 0.1, 1e-5

TAPS3X3 = [(dh, dw) for dh in (-1, 0, 1) for dw in (-1, 0, 1)]   # kernel index kh*3+kw -> offset (kh-1, kw-1)
TAPS1X1 = [(0, 0)]


# When set to a dict, the dense-contraction wrappers bracket each launch with HIP events on the launch stream:
# {kernel instantiation name: [(start_event, end_event, algorithmic_flops), ...]} (used by bench.py only).
PROFILE = None


# Microseconds the launch stream is kept busy ahead of each bracketed launch: the host then has the first event, the
# launch and the second event enqueued before the GPU reaches them, and the pair times the kernel, not the launch gap.
PROFILE_DELAY_US = 40


def _prof_begin(kind: str, flops: float):
    if PROFILE is None:
        return None
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    PROFILE.setdefault(kind, []).append((s, e, flops))
    if PROFILE_DELAY_US:
        _lib.call("mmsa_spin", int(PROFILE_DELAY_US), _stream())
    s.record()
    return e


def _igemm_name(bm, bn, stats, accum, geom, kchunks, nproblems=1, bnrelu=False, offsets=None) -> str:
    """Name of the kernel instantiation mmsa_conv_fwd launches for this tile (as rocprofv3 prints it, without spaces);
    kchunks = 32-channel chunks in a tile's K loop; offsets = the launch's (dh, dw) table."""
    wm, wn = (4, 1) if bn == 32 else (2, 2)
    if offsets is not None and _lib.query("mmsa_conv_shares_tap_rows", *geom, len(offsets), _iarr([o[0] for o in offsets]),
                                          _iarr([o[1] for o in offsets])):
        return "conv_igemm3_kernel<%d,%d,%d,%d,%d,%s,%s>" % (bm, bn, wm, wn, 1 if stats else 0,
                                                              "true" if accum else "false",
                                                              "true" if bnrelu else "false")
    ring = _lib.query("mmsa_conv_ring_buffers", *geom, nproblems, kchunks)
    return "conv_igemm_kernel<%d,%d,%d,%d,%s,%s,%d,%s>" % (bm, bn, wm, wn, "true" if stats else "false",
                                                            "true" if accum else "false", ring,
                                                            "true" if bnrelu else "false")


# the raw handle of the current stream without building a torch.cuda.Stream object around it (this is asked once per
# launch: a launch sequence issued eagerly spends more time here than in the launches otherwise)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


# device of the operands of the launch that is being put together (per thread: autograd runs backward passes on
# threads of its own)
_OPERANDS = threading.local()


def _stream() -> int:
    """Raw handle of the current stream -- of the CURRENT device, which is also where the runtime launches.  Operands
    on another device (a model moved to cuda:1 in a process whose current device is still cuda:0) would be handed to a
    kernel on the wrong GPU: refused here, by name, instead of faulting there."""
    cur = torch.cuda.current_device()
    dev = getattr(_OPERANDS, "dev", cur)
    if dev != cur:
        raise _lib.MmsaError(f"operands live on cuda:{dev} but the current device is cuda:{cur}: call "
                             f"torch.cuda.set_device({dev}) (one process per GPU) or wrap the call in "
                             f"torch.cuda.device({dev})")
    if _RAW_STREAM is not None:
        return _RAW_STREAM(cur)
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=F32, name: str = "tensor") -> int:
    if not t.is_cuda:
        raise _lib.MmsaError(f"{name} must live on the GPU (the HIP kernels have no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    _OPERANDS.dev = t.get_device()
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], dtype=F32, name: str = "tensor"):
    return None if t is None else _chk(t, dtype, name)


def _iarr(vals: Sequence[int]):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _parr(ptrs: Sequence[int]):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


# ---------------------------------------------------------------------------------------------- front end
def logmel(wave: torch.Tensor, tables, out: torch.Tensor, n_target: int = 66150, top_db: float = 80.0,
           specmask: Optional[torch.Tensor] = None):
    """wave [B, n] -> out [B, 1, 64, 1 + n_target // 512]; specmask [B, 4] int32 (first row, rows, first frame, frames)
    zero-fills the dB map before normalisation."""
    if wave.dim() != 2:
        raise ValueError("wave must be [B, n_samples]")
    B, n = wave.shape
    frames = 1 + n_target // 512
    if tuple(out.shape) != (B, 1, 64, frames):
        raise ValueError(f"out must be [{B},1,64,{frames}]")
    if specmask is not None and tuple(specmask.shape) != (B, 4):
        raise ValueError("specmask must be [B, 4]")
    if wave.stride(1) != 1:
        raise ValueError("wave rows must be contiguous")
    if not wave.is_cuda or wave.dtype != F32:
        raise _lib.MmsaError("wave must be a float32 GPU tensor")
    _lib.call("mmsa_logmel_fwd", wave.data_ptr(), wave.stride(0), n, n_target, B, _chk(tables["window"]),
              _chk(tables["tw512"]), _chk(tables["tw1024"]), _chk(tables["melfb"]),
              _chk(tables["melrange"], torch.int32), top_db, _opt(specmask, torch.int32, "specmask"), _chk(out),
              _stream())
    return out


def augment_wave(wave: torch.Tensor, params: torch.Tensor, out: torch.Tensor, noise_factor: float = 0.005,
                 seed: int = 0, lens: torch.Tensor = None):
    """wave [B, n] + params [B, 4] float64 (shift mode, shift, noise on, index step) -> out [B, n_target].
    ``lens`` [B] int32 (optional): valid samples per row, each row augmented as a signal of its own length."""
    if lens is not None and lens.numel() != wave.shape[0]:
        raise ValueError("lens must hold one length per row")
    if wave.dim() != 2 or out.dim() != 2 or wave.shape[0] != out.shape[0] or tuple(params.shape) != (wave.shape[0], 4):
        raise ValueError("wave [B, n], params [B, 4], out [B, n_target]")
    if not wave.is_cuda or wave.dtype != F32 or wave.stride(1) != 1:
        raise _lib.MmsaError("wave must be a float32 GPU tensor with contiguous rows")
    _lib.call("mmsa_augment_wave", wave.data_ptr(), wave.stride(0), wave.shape[1], _opt(lens, torch.int32, "lens"),
              _chk(params, torch.float64, "params"), noise_factor, seed & ((1 << 64) - 1), _chk(out, name="out"), out.shape[1], wave.shape[0], _stream())
    return out


def resample_filters(orig_freq: int, new_freq: int, device, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Filter bank of the sinc resampler as a dict for :func:`resample` (built on the host by the library)."""
    geo = (ctypes.c_int * 4)()
    ptrs = [ctypes.cast(ctypes.byref(geo, 4 * i), ctypes.c_void_p) for i in range(4)]
    _lib.call("mmsa_resample_filters_host", orig_freq, new_freq, lowpass_filter_width, rolloff, None, *ptrs)
    orig, new, width, K = (int(v) for v in geo)
    host = torch.empty((K, (new + 3) // 4 * 4), dtype=F32)      # rows padded to a multiple of 4 floats
    _lib.call("mmsa_resample_filters_host", orig_freq, new_freq, lowpass_filter_width, rolloff, host.data_ptr(), *ptrs)
    return {"kernT": host.to(device), "orig": orig, "new": new, "width": width, "K": K}


def resample(pcm: torch.Tensor, filt, out: torch.Tensor):
    """pcm [B, C, T] at the original rate -> out [B, n_out]: mono mix, sinc resampling, zero pad / trim to n_out."""
    if pcm.dim() != 3 or out.dim() != 2 or out.shape[0] != pcm.shape[0]:
        raise ValueError("pcm must be [B, C, T] and out [B, n_out]")
    B, C, T = pcm.shape
    _lib.call("mmsa_resample_fwd", _chk(pcm, name="pcm"), C, T, C * T, _chk(filt["kernT"]), filt["orig"], filt["new"],
              filt["width"], filt["K"], _chk(out, name="out"), out.shape[1], B, _stream())
    return out


def denoise_filters(sr: int, device, time_constant_s: float = 2.0, freq_mask_smooth_hz: float = 500.0,
                    time_mask_smooth_ms: float = 50.0):
    """Constants of the spectral gate for sample rate ``sr`` (n_fft 1024, hop 256): one-pole coefficient of the
    time smoothing and the normalised triangular taps of the mask-smoothing filter (frequency, time)."""
    t_frames = time_constant_s * sr / 256.0
    b = (math.sqrt(1.0 + 4.0 * t_frames ** 2) - 1.0) / (2.0 * t_frames ** 2)
    n_f = int(freq_mask_smooth_hz / (sr / 512.0))
    n_t = int(time_mask_smooth_ms / ((256.0 / sr) * 1000.0))
    if n_f < 1 or n_t < 1 or 2 * n_f + 1 > 65 or 2 * n_t + 1 > 33:
        raise ValueError(f"mask smoothing of {n_f} bins x {n_t} frames is outside the kernel's range")

    def ramp(n):
        up = torch.arange(1, n + 2, dtype=torch.float64)
        v = torch.cat([up, up[:-1].flip(0)])
        return (v / v.sum()).to(F32).to(device)

    return {"b": b, "vf": ramp(n_f), "vt": ramp(n_t)}


def denoise(wave: torch.Tensor, filt, tables, out: Optional[torch.Tensor] = None, lens: Optional[torch.Tensor] = None,
            thresh: float = 2.0, slope: float = 10.0, prop: float = 1.0, pad: int = 30000, chunk: int = 600000):
    """wave [B, T] -> out [B, T]: non-stationary spectral gating, chunk by chunk for rows longer than ``chunk``.
    ``lens`` [B] int32: valid lengths of rows that hold utterances of different lengths (T <= chunk then); the output
    is zero behind each row's length."""
    if wave.dim() != 2:
        raise ValueError("wave must be [B, T]")
    if not wave.is_cuda or wave.dtype != F32 or wave.stride(1) != 1:
        raise _lib.MmsaError("wave must be a float32 GPU tensor with contiguous rows")
    B, T = wave.shape
    if out is None:
        out = torch.empty((B, T), dtype=F32, device=wave.device)
    if tuple(out.shape) != (B, T) or out.data_ptr() == wave.data_ptr():
        raise ValueError("out must be a separate [B, T] tensor")
    _chk(out, name="out")
    if lens is not None:
        if tuple(lens.shape) != (B,) or T > chunk:
            raise ValueError("lens must be [B] and rows no longer than one chunk")
        _chk(lens, torch.int32, "lens")
    for c0 in range(0, T, chunk):
        L = min(chunk, T - c0)
        geo = (ctypes.c_int * 3)()
        need = ctypes.c_long()
        _lib.call("mmsa_denoise_plan", L, pad, *[ctypes.cast(ctypes.byref(geo, 4 * i), ctypes.c_void_p) for i in range(3)],
                  ctypes.cast(ctypes.byref(need), ctypes.c_void_p))
        ws = torch.empty((B * need.value,), dtype=F32, device=wave.device)
        _lib.call("mmsa_denoise_fwd", wave.data_ptr(), wave.stride(0), T, c0, L, pad, B,
                  None if lens is None else lens.data_ptr(), _chk(tables["window"]),
                  _chk(tables["tw512"]), _chk(tables["tw1024"]), _chk(filt["vt"]), filt["vt"].numel(), _chk(filt["vf"]),
                  filt["vf"].numel(), filt["b"], thresh, slope, prop, ws.data_ptr(), out.data_ptr(), out.stride(0),
                  _stream())
    return out


# ---------------------------------------------------------------------------------------------- convolution
def pack_weights(w: torch.Tensor, fwd: Optional[torch.Tensor], dgrad: Optional[torch.Tensor]):
    cout, cin = w.shape[0], w.shape[1]
    ntaps = w.numel() // (cout * cin)
    for t in (fwd, dgrad):
        if t is not None and t.numel() != w.numel():
            raise ValueError("packed buffer size mismatch")
    _lib.call("mmsa_conv_pack_weights", _chk(w), _opt(fwd), _opt(dgrad), cout, cin, ntaps, _stream())


class PackPlan:
    """A device-resident table of (weight, forward pack, data-gradient pack) triples: one launch re-packs them all.
    A table is built once per distinct set of pointers and kept for the plan's lifetime (weights and workspaces are
    persistent in training): a captured launch sequence holds the table's address, so a table must neither move nor
    change when a batch of another shape passes through the same plan."""

    def __init__(self):
        self.tables = {}

    def run(self, triples):
        import numpy as np
        key = tuple((w.data_ptr(), f.data_ptr() if f is not None else 0, d.data_ptr() if d is not None else 0)
                    for w, f, d in triples)
        table = self.tables.get(key)
        if table is None:
            rec = _lib.query("mmsa_conv_pack_desc_bytes")
            dt = np.dtype([("w", "<u8"), ("fwd", "<u8"), ("dgrad", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"),
                           ("ntaps", "<i4"), ("pad", "<i4")])
            if dt.itemsize != rec:
                raise _lib.MmsaError("pack descriptor layout mismatch")
            arr = np.zeros(len(triples), dtype=dt)
            for i, (w, f, d) in enumerate(triples):
                _chk(w)
                cout, cin = w.shape[0], w.shape[1]
                ntaps = w.numel() // (cout * cin)
                if not 1 <= ntaps <= 9:
                    raise ValueError("pack: a convolution weight has 1 to 9 taps")
                for x in (f, d):
                    if x is not None and (x.numel() != w.numel() or _chk(x) == 0):
                        raise ValueError("packed buffer size mismatch")
                arr[i] = (key[i][0], key[i][1], key[i][2], cout, cin, ntaps, 0)
            table = self.tables[key] = torch.from_numpy(arr.view(np.uint8)).to(triples[0][0].device)
        _lib.call("mmsa_conv_pack_weights_multi", table.data_ptr(), len(triples), _stream())


def conv_tile_rows(B, H, W, Cout) -> int:
    return _lib.query("mmsa_conv_tile_rows", B, H, W, Cout)


def conv_stats_rows(B, H, W, Cout):
    bm = conv_tile_rows(B, H, W, Cout)
    tiles = (H * W + bm - 1) // bm
    return B * tiles, tiles


def conv_fwd(X, Wp, Y, stats, B, H, W, Cin, Cout, taps, negate=False, accumulate=False):
    if X.numel() != B * H * W * Cin or Y.numel() != B * H * W * Cout or Wp.numel() != len(taps) * Cin * Cout:
        raise ValueError("conv_fwd: operand sizes do not match the geometry")
    if stats is not None:
        rows, _ = conv_stats_rows(B, H, W, Cout)
        if stats.numel() != rows * 2 * Cout:
            raise ValueError("conv_fwd: stats buffer has the wrong size")
    sgn = -1 if negate else 1
    ev = None
    if PROFILE is not None:
        bm, bn = (128, 32) if Cout % 64 else (conv_tile_rows(B, H, W, Cout), 64)
        kind = _igemm_name(bm, bn, stats is not None, accumulate, (B, H, W, Cout), len(taps) * Cin // 32,
                           offsets=[(sgn * t[0], sgn * t[1]) for t in taps])
        ev = _prof_begin(kind, 2.0 * B * H * W * Cin * Cout * len(taps))
    _lib.call("mmsa_conv_fwd", _chk(X), _chk(Wp), _chk(Y), _opt(stats), B, H, W, Cin, Cout, len(taps),
              _iarr([sgn * t[0] for t in taps]), _iarr([sgn * t[1] for t in taps]), int(accumulate), _stream())
    if ev is not None:
        ev.record()


def conv_fwd_dual(X, Wp, X2, Wp2, Y, B, H, W, Cin, Cout, taps, negate=False):
    """Y = conv(X, Wp) + (1x1) X2 . Wp2 in one launch; X2 has the channel count of X."""
    if (X.numel() != B * H * W * Cin or X2.numel() != X.numel() or Y.numel() != B * H * W * Cout
            or Wp.numel() != len(taps) * Cin * Cout or Wp2.numel() != Cin * Cout):
        raise ValueError("conv_fwd_dual: operand sizes do not match the geometry")
    sgn = -1 if negate else 1
    ev = None
    if PROFILE is not None:
        bm, bn = (128, 32) if Cout % 64 else (conv_tile_rows(B, H, W, Cout), 64)
        ev = _prof_begin(_igemm_name(bm, bn, False, False, (B, H, W, Cout), (len(taps) + 1) * Cin // 32,
                                     offsets=[(sgn * t[0], sgn * t[1]) for t in taps]),
                         2.0 * B * H * W * Cin * Cout * (len(taps) + 1))
    _lib.call("mmsa_conv_fwd_dual", _chk(X), _chk(Wp), _chk(X2), _chk(Wp2), _chk(Y), B, H, W, Cin, Cout, len(taps),
              _iarr([sgn * t[0] for t in taps]), _iarr([sgn * t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()


def conv_dual_prefers_split(B, H, W, Cout) -> bool:
    """Whether :func:`conv_fwd_dual` should run with its K range split in two for this geometry (few output tiles)."""
    return bool(_lib.query("mmsa_conv_fwd_dual_prefers_split", B, H, W, Cout))


def conv_fwd_dual_split(X, Wp, X2, Wp2, Y, B, H, W, Cin, Cout, taps, negate=False):
    """:func:`conv_fwd_dual` with two workgroups per output tile adding into ``Y``, which must be zero on entry."""
    if (X.numel() != B * H * W * Cin or X2.numel() != X.numel() or Y.numel() != B * H * W * Cout
            or Wp.numel() != len(taps) * Cin * Cout or Wp2.numel() != Cin * Cout or len(taps) != 9):
        raise ValueError("conv_fwd_dual_split: operand sizes do not match the geometry")
    sgn = -1 if negate else 1
    ev = None
    if PROFILE is not None:
        ev = _prof_begin(_igemm_name(64, 64, False, True, (B, H, W, Cout), (len(taps) + 1) * Cin // 32, 2,
                                     offsets=[(sgn * t[0], sgn * t[1]) for t in taps]),
                         2.0 * B * H * W * Cin * Cout * (len(taps) + 1))
    _lib.call("mmsa_conv_fwd_dual_split", _chk(X), _chk(Wp), _chk(X2), _chk(Wp2), _chk(Y), B, H, W, Cin, Cout,
              len(taps), _iarr([sgn * t[0] for t in taps]), _iarr([sgn * t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()


def conv_fwd_multi(X, Wps, Ys, ntaps_list, B, H, W, Cin, Cout, taps, tap0_list=None, stats_list=None,
                   bnrelu_list=None):
    """Several convolutions of X in one launch; problem i uses offsets tap0_list[i] .. + ntaps_list[i] of ``taps``
    (from 0 when tap0_list is None) and, with stats_list, writes its per-tile statistics like :func:`conv_fwd`.
    bnrelu_list: per problem None or a [4][Cout] BatchNorm coefficient table; that problem then stores
    relu(scale * conv + shift) (inference: the table does not depend on the convolution)."""
    n = len(Wps)
    if X.numel() != B * H * W * Cin or len(Ys) != n or len(ntaps_list) != n or n > 4:
        raise ValueError("conv_fwd_multi: bad arguments")
    tap0_list = [0] * n if tap0_list is None else list(tap0_list)
    for wp, y, nt, t0 in zip(Wps, Ys, ntaps_list, tap0_list):
        if wp.numel() != nt * Cin * Cout or y.numel() != B * H * W * Cout or t0 < 0 or t0 + nt > len(taps):
            raise ValueError("conv_fwd_multi: operand sizes do not match the geometry")
    bm = 64
    if bnrelu_list is not None and (len(bnrelu_list) != n or any(c is not None and c.numel() != 4 * Cout
                                                                 for c in bnrelu_list)):
        raise ValueError("conv_fwd_multi: BatchNorm tables have the wrong size")
    if stats_list is not None:
        rows, _ = conv_stats_rows(B, H, W, Cout)
        bm = _lib.query("mmsa_conv_tile_rows", B, H, W, Cout)
        if len(stats_list) != n or any(st.numel() != rows * 2 * Cout for st in stats_list):
            raise ValueError("conv_fwd_multi: statistics buffers have the wrong size")
    ev = None
    if PROFILE is not None:
        ev = _prof_begin(_igemm_name(bm, 64, stats_list is not None, False, (B, H, W, Cout), len(taps) * Cin // 32, n,
                                     bnrelu_list is not None and any(c is not None for c in bnrelu_list),
                                     offsets=list(taps) if all(
                                         (nt, t0) in ((9, 0), (1, 4)) for nt, t0 in zip(ntaps_list, tap0_list)) else None),
                         2.0 * B * H * W * Cin * Cout * sum(ntaps_list))
    _lib.call("mmsa_conv_fwd_multi", _chk(X), _parr([_chk(w) for w in Wps]), _parr([_chk(y) for y in Ys]),
              _iarr(ntaps_list), _iarr(tap0_list), None if stats_list is None else _parr([_chk(t) for t in stats_list]),
              None if bnrelu_list is None else _parr([_opt(c) for c in bnrelu_list]), n, B, H, W, Cin, Cout, len(taps), _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()


def conv_wgrad_workspace(B, H, W, Cin, Cout, ntaps) -> int:
    return _lib.query("mmsa_conv_wgrad_workspace_floats", B, H, W, Cin, Cout, ntaps, None)


def conv_wgrad_supported(B, H, W, Cin, Cout) -> bool:
    """False when the map is too wide for any weight-gradient kernel (a one-row strip with its halo must fit in LDS:
    about 238 columns for the 3x3 convolutions)."""
    return _lib.load().mmsa_conv_wgrad_workspace_floats(B, H, W, Cin, Cout, 9, None) != -3


def conv_wgrad(X, dY, dW, workspace, B, H, W, Cin, Cout, taps, w_valid=None, accumulate=False):
    if X.numel() != B * H * W * Cin or dY.numel() != B * H * W * Cout or dW.numel() != len(taps) * Cin * Cout:
        raise ValueError("conv_wgrad: operand sizes do not match the geometry")
    if workspace.numel() < conv_wgrad_workspace(B, H, W, Cin, Cout, len(taps)):
        raise ValueError("conv_wgrad: workspace too small")
    ev = _prof_begin("conv_wgrad_kernel<%d>" % len(taps), 2.0 * B * H * W * Cin * Cout * len(taps))
    _lib.call("mmsa_conv_wgrad", _chk(X), _chk(dY), _chk(workspace), B, H, W,
              W if w_valid is None else w_valid, Cin, Cout, len(taps), _iarr([t[0] for t in taps]),
              _iarr([t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()
    _lib.call("mmsa_conv_wgrad_reduce", _chk(workspace), _chk(dW), B, H, W, Cin, Cout, len(taps), int(accumulate),
              _stream())


def conv_wgrad_pair(X, dY9, dW9, dY1, dW1, ws9, ws1, B, H, W, Cin, Cout, taps=None):
    """dW9 (3x3, from dY9) and dW1 (1x1, from dY1) of two convolutions of the same input X: one launch + one reduction."""
    taps = TAPS3X3 if taps is None else taps
    n = B * H * W
    if X.numel() != n * Cin or dY9.numel() != n * Cout or dY1.numel() != n * Cout or dW9.numel() != 9 * Cin * Cout \
            or dW1.numel() != Cin * Cout or len(taps) != 9:
        raise ValueError("conv_wgrad_pair: operand sizes do not match the geometry")
    if ws9.numel() < conv_wgrad_workspace(B, H, W, Cin, Cout, 9) or ws1.numel() < conv_wgrad_workspace(B, H, W, Cin, Cout, 1):
        raise ValueError("conv_wgrad_pair: workspace too small")
    ev = _prof_begin("conv_wgrad_pair_kernel", 2.0 * n * Cin * Cout * 10)
    _lib.call("mmsa_conv_wgrad_pair", _chk(X), _chk(dY9), _chk(dY1), _chk(ws9), _chk(ws1), _chk(dW9), _chk(dW1), B, H, W,
              Cin, Cout, _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()


def conv_wgrad_slab_count(B, H, W, Cin, Cout, with_shortcut: bool) -> int:
    """Partial slabs one launch of :func:`conv_wgrad_slabs` writes for this geometry; 0 if two one-row strips of the map
    do not fit in LDS (rows wider than about 120 positions, 85 with the shortcut riding along), in which case the
    first-generation kernels serve."""
    n = ctypes.c_int()
    rc = _lib.load().mmsa_conv_wgrad_slab_floats(B, H, W, Cin, Cout, int(with_shortcut),
                                                 ctypes.cast(ctypes.byref(n), ctypes.c_void_p))
    if rc == -3:
        return 0
    if rc < 0:
        raise _lib.MmsaError(f"mmsa_conv_wgrad_slab_floats rejected its arguments (status {rc})")
    return n.value


def conv_wgrad_slabs(X, dY, dY_sc, slabs, slabs_sc, B, H, W, Cin, Cout, taps=None):
    """Partial sums of the 3x3 weight gradient (from dY) and, when dY_sc is given, of the 1x1 shortcut's (from dY_sc;
    both convolutions read X): slabs [nslabs, 9, Cin, Cout], slabs_sc [nslabs, Cin, Cout].  :class:`ReducePlan` turns
    slabs into gradients."""
    taps = TAPS3X3 if taps is None else taps
    n = B * H * W
    if X.numel() != n * Cin or dY.numel() != n * Cout or len(taps) != 9 or (dY_sc is not None and dY_sc.numel() != n * Cout):
        raise ValueError("conv_wgrad_slabs: operand sizes do not match the geometry")
    ns = conv_wgrad_slab_count(B, H, W, Cin, Cout, dY_sc is not None)
    if slabs.numel() < ns * 9 * Cin * Cout or (dY_sc is not None and (slabs_sc is None or slabs_sc.numel() < ns * Cin * Cout)):
        raise ValueError("conv_wgrad_slabs: slab buffer too small")
    kind = "conv_wgrad2_kernel<%s,%d>" % ("true" if dY_sc is not None else "false", W if W in (16, 32, 65) else 0)
    ev = _prof_begin(kind, 2.0 * n * Cin * Cout * (10 if dY_sc is not None else 9))
    _lib.call("mmsa_conv_wgrad_slabs", _chk(X), _chk(dY), _opt(dY_sc), _chk(slabs), _opt(slabs_sc), B, H, W, Cin, Cout,
              _iarr([t[0] for t in taps]), _iarr([t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()
    return ns


class ReducePlan:
    """A device-resident table of (slabs, dW, nslabs, ntaps, Cin, Cout) records: one launch sums the slabs of all of
    them into the reference's [Cout][Cin][kh][kw] gradients.  A table is built once per distinct set of entries and kept
    for the plan's lifetime (see PackPlan)."""

    def __init__(self):
        self.tables = {}

    def run(self, entries):
        import numpy as np
        key = tuple((s.data_ptr(), d.data_ptr(), int(ns), int(nt), int(ci), int(co)) for s, d, ns, nt, ci, co in entries)
        hit = self.tables.get(key)
        if hit is None:
            dt = np.dtype([("slabs", "<u8"), ("dW", "<u8"), ("nslabs", "<i4"), ("ntaps", "<i4"), ("Cin", "<i4"),
                           ("Cout", "<i4"), ("block0", "<i4"), ("nblocks", "<i4")])
            if dt.itemsize != _lib.query("mmsa_conv_wgrad_reduce_desc_bytes"):
                raise _lib.MmsaError("reduce descriptor layout mismatch")
            arr = np.zeros(len(entries), dtype=dt)
            b0 = 0
            for i, (s, d, ns, nt, ci, co) in enumerate(entries):
                _chk(s), _chk(d)
                if d.numel() != nt * ci * co or s.numel() < ns * nt * ci * co or ns < 1:
                    raise ValueError("ReducePlan: slab / gradient sizes do not match")
                nb = (nt * ci * co + 255) // 256
                arr[i] = (key[i][0], key[i][1], ns, nt, ci, co, b0, nb)
                b0 += nb
            hit = self.tables[key] = (torch.from_numpy(arr.view(np.uint8)).to(entries[0][0].device), b0)
        _lib.call("mmsa_conv_wgrad_reduce_multi", hit[0].data_ptr(), len(entries), hit[1], _stream())


# ---------------------------------------------------------------------------------------------- audio branch
def stem_stats_rows(B, H) -> int:
    """Rows of the stem's statistics table (one per group of image rows): ``partial`` holds 64 floats per row."""
    return _lib.query("mmsa_stem_stats_rows", B, H)


def stem_stats(x, w, bias, partial, B, H, W):
    if (x.numel() != B * H * W or partial.numel() != stem_stats_rows(B, H) * 64 or w.numel() != 288
            or bias.numel() != 32):
        raise ValueError("stem_stats: bad sizes")
    _lib.call("mmsa_stem_stats", _chk(x), _chk(w), _chk(bias), _chk(partial), B, H, W, _stream())


def stem_apply_pool(x, w, bias, coeff, out, B, H, W):
    if x.numel() != B * H * W or out.numel() != B * (H // 2) * (W // 2) * 32 or coeff.numel() != 128:
        raise ValueError("stem_apply_pool: bad sizes")
    _lib.call("mmsa_stem_apply_pool", _chk(x), _chk(w), _chk(bias), _chk(coeff), _chk(out), B, H, W, _stream())


def stem_bwd_scratch(B, H) -> int:
    return B * (H // 2) * 384 + 64


def stem_bwd(x, w, bias, coeff, dout, dW, dbias, dgamma, dbeta, scratch, B, H, W, training):
    if dout.numel() != B * (H // 2) * (W // 2) * 32 or scratch.numel() < stem_bwd_scratch(B, H):
        raise ValueError("stem_bwd: bad sizes")
    if dW.numel() != 288 or dbias.numel() != 32 or dgamma.numel() != 32 or dbeta.numel() != 32:
        raise ValueError("stem_bwd: bad gradient sizes")
    _lib.call("mmsa_stem_bwd", _chk(x), _chk(w), _chk(bias), _chk(coeff), _chk(dout), _chk(dW), _chk(dbias),
              _chk(dgamma), _chk(dbeta), _chk(scratch), B, H, W, int(training), _stream())


def bn_finalize(partial, rows, tiles_per_img, count, gamma, beta, rmean, rvar, nbt, training, coeff, bsum, C):
    if coeff.numel() != 4 * C or gamma.numel() != C:
        raise ValueError("bn_finalize: bad sizes")
    if partial is not None and partial.numel() != rows * 2 * C:
        raise ValueError("bn_finalize: partial buffer has the wrong size")
    if bsum is not None and bsum.numel() != (rows // tiles_per_img) * C:
        raise ValueError("bn_finalize: bsum has the wrong size")
    _lib.call("mmsa_bn_finalize", _opt(partial), rows, tiles_per_img, float(count), _chk(gamma), _chk(beta),
              _opt(rmean), _opt(rvar), _opt(nbt, torch.int64), BN_MOMENTUM, BN_EPS, int(training), _chk(coeff),
              _opt(bsum), C, _stream())


def bn_finalize_pair(partials, rows, tiles_per_img, count, bns, training, coeffs, C):
    """:func:`bn_finalize` for two BatchNorm layers (``bns``: (gamma, beta, running_mean, running_var, nbt) each)."""
    if len(partials) != 2 or len(bns) != 2 or len(coeffs) != 2:
        raise ValueError("bn_finalize_pair takes two layers")
    for p_, bn, cf in zip(partials, bns, coeffs):
        if cf.numel() != 4 * C or bn[0].numel() != C or (p_ is not None and p_.numel() != rows * 2 * C):
            raise ValueError("bn_finalize_pair: bad sizes")
    col = lambda k, dt=F32: _parr([0 if bn[k] is None else _chk(bn[k], dt) for bn in bns])   # noqa: E731
    _lib.call("mmsa_bn_finalize_pair", _parr([0 if p_ is None else _chk(p_) for p_ in partials]), rows, tiles_per_img,
              float(count), col(0), col(1), col(2), col(3), col(4, torch.int64), BN_MOMENTUM, BN_EPS, int(training),
              _parr([_chk(c) for c in coeffs]), C, _stream())


def bn_relu_apply(x, coeff, y, npos, C):
    if x.numel() != npos * C or y.numel() != npos * C or coeff.numel() != 4 * C:
        raise ValueError("bn_relu_apply: bad sizes")
    _lib.call("mmsa_bn_relu_apply", _chk(x), _chk(coeff), _chk(y), npos, C, _stream())


def se_gate(bsum, coeff, W1, W2, HW, gap, hid, gate, B, C):
    Ch = W1.shape[0]
    if W1.numel() != Ch * C or W2.numel() != C * Ch or bsum.numel() != B * C or gate.numel() != B * C \
            or gap.numel() != B * C or hid.numel() != B * Ch:
        raise ValueError("se_gate: bad sizes")
    _lib.call("mmsa_se_gate", _chk(bsum), _chk(coeff), _chk(W1), _chk(W2), HW, _chk(gap), _chk(hid), _chk(gate),
              B, C, Ch, _stream())


def res_gate_relu_pool(c2, sc, coeff2, coeffs, gate, out, B, H, W, C):
    n = B * H * W * C
    if c2.numel() != n or sc.numel() != n or out.numel() != B * (H // 2) * (W // 2) * C or gate.numel() != B * C:
        raise ValueError("res_gate_relu_pool: bad sizes")
    _lib.call("mmsa_res_gate_relu_pool", _chk(c2), _chk(sc), _chk(coeff2), _chk(coeffs), _chk(gate), _chk(out),
              B, H, W, C, _stream())


def res_gate_relu_pool_mean(c2, sc, coeff2, coeffs, gate, feat, B, H, W, C):
    """Mean over the pooled positions of what :func:`res_gate_relu_pool` would store: feat [B, C]."""
    n = B * H * W * C
    if c2.numel() != n or sc.numel() != n or feat.numel() != B * C or gate.numel() != B * C:
        raise ValueError("res_gate_relu_pool_mean: bad sizes")
    _lib.call("mmsa_res_gate_relu_pool_mean", _chk(c2), _chk(sc), _chk(coeff2), _chk(coeffs), _chk(gate), _chk(feat),
              B, H, W, C, _stream())


def res_se_gate_relu_pool(c2, sc, coeff2, coeffs, bsum, W1, W2, gap, hid, gate, out, feat, B, H, W, C):
    """:func:`se_gate` and the pooling that consumes its gate in one launch: into ``out`` [B, H/2, W/2, C]
    (:func:`res_gate_relu_pool`) or, with ``out`` None, into ``feat`` [B, C] (:func:`res_gate_relu_pool_mean`)."""
    Ch = W1.shape[0]
    n = B * H * W * C
    if (c2.numel() != n or sc.numel() != n or W1.numel() != Ch * C or W2.numel() != C * Ch or bsum.numel() != B * C
            or gate.numel() != B * C or gap.numel() != B * C or hid.numel() != B * Ch or (out is None) == (feat is None)
            or (out is not None and out.numel() != B * (H // 2) * (W // 2) * C)
            or (feat is not None and feat.numel() != B * C)):
        raise ValueError("res_se_gate_relu_pool: bad sizes")
    _lib.call("mmsa_res_se_gate_relu_pool", _chk(c2), _chk(sc), _chk(coeff2), _chk(coeffs), _chk(bsum), _chk(W1),
              _chk(W2), _chk(gap), _chk(hid), _chk(gate), _opt(out), _opt(feat), B, H, W, C, Ch, _stream())


def gate_affine_add(a, coeff2, gate, sc, coeffs, out, B, P, C, relu: bool):
    """out = [relu]((a * scale2 + shift2) * gate + (sc * scale_s + shift_s)) over [B, P, C] maps; coefficient tables,
    gate and second map are optional (identity / 1 / 0)."""
    n = B * P * C
    if a.numel() != n or out.numel() != n or (sc is not None and sc.numel() != n) or C % 4:
        raise ValueError("gate_affine_add: bad sizes")
    for t in (coeff2, coeffs):
        if t is not None and t.numel() != 4 * C:
            raise ValueError("gate_affine_add: coefficient tables are [4, C]")
    if gate is not None and gate.numel() != B * C:
        raise ValueError("gate_affine_add: gate is [B, C]")
    if coeffs is not None and sc is None:
        raise ValueError("gate_affine_add: coeffs without a second map")
    _lib.call("mmsa_gate_affine_add", _chk(a), _opt(coeff2), _opt(gate), _opt(sc), _opt(coeffs), _chk(out), B, P, C,
              int(relu), _stream())


def gap_fwd(x, out, B, P, C):
    if x.numel() != B * P * C or out.numel() != B * C:
        raise ValueError("gap_fwd: bad sizes")
    _lib.call("mmsa_gap_fwd", _chk(x), _chk(out), B, P, C, _stream())


def gap_bwd(dout, dx, B, P, C):
    if dx.numel() != B * P * C or dout.numel() != B * C:
        raise ValueError("gap_bwd: bad sizes")
    _lib.call("mmsa_gap_bwd", _chk(dout), _chk(dx), B, P, C, _stream())


def resblock_bwd_scratch(B, H, W, C, unpooled=False) -> int:
    return _lib.query("mmsa_resblock_bwd_unpooled_scratch_floats" if unpooled else "mmsa_resblock_bwd_scratch_floats",
                      B, H, W, C)


def resblock_bwd_ew(dout, c2, sc, coeff2, coeffs, gate, hid, gap, bsum, W1, W2, d_c2, d_sc, dgamma2, dbeta2,
                    dgammas, dbetas, dW1, dW2, scratch, B, H, W, C, training, dout_is_mean=False, unpooled=False):
    """dout_is_mean: dout is [B, C], the gradient with respect to the mean of the block's pooled output.
    unpooled: no pooling follows the block, dout is [B, H, W, C]; ``training`` may then carry bit 1 (= 2): the shortcut
    is a fixed affine map (the identity shortcut, with coeffs = [0, 1, 1, 0] per channel)."""
    n = B * H * W * C
    want = n if unpooled else (B * C if dout_is_mean else B * (H // 2) * (W // 2) * C)
    if dout.numel() != want or c2.numel() != n or sc.numel() != n or d_c2.numel() != n \
            or d_sc.numel() != n or scratch.numel() < resblock_bwd_scratch(B, H, W, C, unpooled):
        raise ValueError("resblock_bwd_ew: bad sizes")
    for t in (dgamma2, dbeta2, dgammas, dbetas):
        if t.numel() != C:
            raise ValueError("resblock_bwd_ew: bad BN gradient size")
    if (dW1 is None) != (dW2 is None):
        raise ValueError("resblock_bwd_ew: pass both SE gradients or neither")
    if dW1 is not None and (dW1.numel() != W1.numel() or dW2.numel() != W2.numel()):
        raise ValueError("resblock_bwd_ew: bad SE gradient size")
    entry = ("mmsa_resblock_bwd_ew_unpooled" if unpooled else
             "mmsa_resblock_bwd_ew_from_mean" if dout_is_mean else "mmsa_resblock_bwd_ew")
    _lib.call(entry, _chk(dout), _chk(c2), _chk(sc), _chk(coeff2), _chk(coeffs), _chk(gate),
              _chk(hid), _chk(gap), _chk(bsum), _chk(W1), _chk(W2), _chk(d_c2), _chk(d_sc), _chk(dgamma2),
              _chk(dbeta2), _chk(dgammas), _chk(dbetas), _opt(dW1), _opt(dW2), _chk(scratch), B, H, W, C,
              int(training), _stream())


def resblock_bwd_se_wgrad(scratch, hid, gap, dW1, dW2, B, H, W, C):
    """The squeeze-excite weight gradients left out of :func:`resblock_bwd_ew` when it was called with dW1 = dW2 = None."""
    if scratch.numel() < resblock_bwd_scratch(B, H, W, C) or dW1.numel() != dW2.numel() or dW1.numel() != C * (C // 16):
        raise ValueError("resblock_bwd_se_wgrad: bad sizes")
    _lib.call("mmsa_resblock_bwd_se_wgrad", _chk(scratch), _chk(hid), _chk(gap), _chk(dW1), _chk(dW2), B, H, W, C,
              _stream())


def copy_multi(pairs):
    """[(dst, src), ...] (at most four; same dtype, shape and device, contiguous) copied in one launch."""
    n = len(pairs)
    if not 1 <= n <= 4:
        raise ValueError("copy_multi takes one to four (dst, src) pairs")
    srcs, dsts, sizes = [], [], []
    for dst, src in pairs:
        if dst.shape != src.shape or dst.dtype != src.dtype or not (dst.is_cuda and src.is_cuda) \
                or not (dst.is_contiguous() and src.is_contiguous()):
            raise ValueError("copy_multi: pairs must match in shape / dtype and be contiguous GPU tensors")
        srcs.append(src.data_ptr())
        dsts.append(dst.data_ptr())
        sizes.append(dst.numel() * dst.element_size())
    _lib.call("mmsa_copy_multi", (ctypes.c_void_p * n)(*srcs), (ctypes.c_void_p * n)(*dsts), (ctypes.c_long * n)(*sizes),
              n, _stream())


def bn_relu_bwd_scratch(npos, C) -> int:
    return _lib.query("mmsa_bn_relu_bwd_scratch_floats", npos, C)


def bn_relu_bwd(dh, c1, coeff, dgamma, dbeta, scratch, npos, C, training):
    if dh.numel() != npos * C or c1.numel() != npos * C or scratch.numel() < bn_relu_bwd_scratch(npos, C):
        raise ValueError("bn_relu_bwd: bad sizes")
    _lib.call("mmsa_bn_relu_bwd", _chk(dh), _chk(c1), _chk(coeff), _chk(dgamma), _chk(dbeta), _chk(scratch), npos, C,
              int(training), _stream())


def conv_shares_tap_rows(B, H, W, Cout, taps, negate=False) -> bool:
    """Whether a convolution with this offset table runs on the kernel that stages an input strip once per tap row."""
    sgn = -1 if negate else 1
    return bool(_lib.query("mmsa_conv_shares_tap_rows", B, H, W, Cout, len(taps), _iarr([sgn * t[0] for t in taps]),
                           _iarr([sgn * t[1] for t in taps])))


def conv_fwd_bnrelu_bwd_stats(X, Wp, Y, stats, x, coeff, B, H, W, Cin, Cout, taps, negate=False):
    """:func:`conv_fwd` for a data gradient Y = d loss / d relu(bn(x)): the launch also leaves the per-tile sums of
    g = Y [bn(x) > 0] and g * xhat in ``stats`` ([2][Cout][rows], rows as :func:`conv_stats_rows`)."""
    rows, _ = conv_stats_rows(B, H, W, Cout)
    if (X.numel() != B * H * W * Cin or Y.numel() != B * H * W * Cout or Wp.numel() != len(taps) * Cin * Cout
            or x.numel() != Y.numel() or coeff.numel() != 4 * Cout or stats.numel() != rows * 2 * Cout):
        raise ValueError("conv_fwd_bnrelu_bwd_stats: operand sizes do not match the geometry")
    sgn = -1 if negate else 1
    ev = None
    if PROFILE is not None:
        ev = _prof_begin("conv_igemm3_kernel<64,64,2,2,2,false,false>", 2.0 * B * H * W * Cin * Cout * len(taps))
    _lib.call("mmsa_conv_fwd_bnrelu_bwd_stats", _chk(X), _chk(Wp), _chk(Y), _chk(stats), _chk(x), _chk(coeff), B, H, W,
              Cin, Cout, len(taps), _iarr([sgn * t[0] for t in taps]), _iarr([sgn * t[1] for t in taps]), _stream())
    if ev is not None:
        ev.record()
    return rows


def bn_relu_bwd_finish(dh, c1, coeff, partial, rows, dgamma, dbeta, cb, npos, C, training):
    if dh.numel() != npos * C or c1.numel() != npos * C or partial.numel() != rows * 2 * C or cb.numel() < 3 * C:
        raise ValueError("bn_relu_bwd_finish: bad sizes")
    _lib.call("mmsa_bn_relu_bwd_finish", _chk(dh), _chk(c1), _chk(coeff), _chk(partial), rows, _chk(dgamma), _chk(dbeta),
              _chk(cb), npos, C, int(training), _stream())


# ---------------------------------------------------------------------------------------------- text branch
def embedding_gather(ids, table, out, err=None):
    ntok = ids.numel()
    V, E = table.shape
    if out.numel() != ntok * E:
        raise ValueError("embedding_gather: bad output size")
    _lib.call("mmsa_embedding_gather", _chk(ids, torch.int64, "input_ids"), _chk(table), _chk(out), ntok, E, V,
              _opt(err, torch.int32), _stream())


def bias_relu_maxtime(y, bias, feat, arg, B, L, Lout, F, ldf, col0):
    if y.numel() != B * L * F or feat.shape != (B, ldf) or arg.shape != (B, ldf) or col0 + F > ldf:
        raise ValueError("bias_relu_maxtime: bad sizes")
    _chk(feat), _chk(arg, torch.int32)
    _lib.call("mmsa_bias_relu_maxtime", _chk(y), _chk(bias), feat.data_ptr() + 4 * col0, arg.data_ptr() + 4 * col0,
              B, L, Lout, F, ldf, _stream())


def bias_relu_maxtime_multi(ys, biases, louts, feat, arg, B, L, F):
    n = len(ys)
    if feat.shape != (B, n * F) or arg.shape != (B, n * F) or any(y.numel() != B * L * F for y in ys):
        raise ValueError("bias_relu_maxtime_multi: bad sizes")
    _lib.call("mmsa_bias_relu_maxtime_multi", _parr([_chk(y) for y in ys]), _parr([_chk(b) for b in biases]),
              _iarr(louts), n, _chk(feat), _chk(arg, torch.int32), B, L, F, n * F, _stream())


def token_groups(ids, vocab, padding_idx, groups, id_table):
    """Sorts the token positions of ``ids`` into groups of equal id for :func:`text_bwd` (``groups``: 5 * ids.numel()
    int32; ``id_table``: 2 * vocab int32, zero before the first call and left zero by every call)."""
    n = ids.numel()
    if groups.numel() < 5 * n or id_table.numel() < 2 * vocab:
        raise ValueError("token_groups: workspace too small")
    _lib.call("mmsa_token_groups", _chk(ids, torch.int64, "input_ids"), n, vocab, padding_idx,
              _chk(groups, torch.int32), _chk(id_table, torch.int32), _stream())


def text_bwd(dfeat, feat, arg, emb, ids, ksizes, wps, dWs, dbs, dtable, B, L, E, F, padding_idx=0, err=None,
             drow=None, groups=None):
    """``groups``: what :func:`token_groups` left for these ids (computed here when not given); ``drow``: [B*L, E]
    floats of scratch."""
    n = len(ksizes)
    ldf = n * F
    if dfeat.shape != (B, ldf) or feat.shape != (B, ldf) or arg.shape != (B, ldf) or emb.numel() != B * L * E:
        raise ValueError("text_bwd: bad sizes")
    for k, wp, dW, db in zip(ksizes, wps, dWs, dbs):
        if wp.numel() != k * F * E or dW.numel() != k * F * E or db.numel() != F:
            raise ValueError("text_bwd: bad weight sizes")
    if dtable.dim() != 2 or dtable.shape[1] != E:
        raise ValueError("text_bwd: dtable must be [vocab, E]")
    if drow is None:
        drow = torch.empty(B * L * E, dtype=F32, device=dtable.device)
    if groups is None:
        groups = torch.empty(5 * B * L, dtype=torch.int32, device=dtable.device)
        token_groups(ids, dtable.shape[0], padding_idx, groups,
                     torch.zeros(2 * dtable.shape[0], dtype=torch.int32, device=dtable.device))
    if drow.numel() < B * L * E or groups.numel() < 5 * B * L:
        raise ValueError("text_bwd: workspace too small")
    _lib.call("mmsa_text_bwd", _chk(dfeat), _chk(feat), _chk(arg, torch.int32), _chk(emb),
              _chk(ids, torch.int64), n, _iarr(ksizes), _iarr([i * F for i in range(n)]),
              _parr([_chk(w) for w in wps]), _parr([_chk(w) for w in dWs]), _parr([_chk(w) for w in dbs]),
              _chk(dtable), _chk(drow), _chk(groups, torch.int32), B, L, E, F, ldf,
              padding_idx, dtable.shape[0], _opt(err, torch.int32), _stream())


# ---------------------------------------------------------------------------------------------- dense head
def _rows(t: torch.Tensor, name: str):
    """2-D row view: (pointer, leading dimension); rows must be unit-stride."""
    if t.dim() != 2 or t.stride(1) != 1 or not t.is_cuda or t.dtype != F32:
        raise ValueError(f"{name}: need a float32 GPU matrix with unit-stride rows")
    return t.data_ptr(), t.stride(0)


def dense_fwd(X, W, bias, gamma, beta, Y, pre, stats, epilogue=EPI_NONE, drop_p=0.0, seed=0, step=None):
    B, K = X.shape
    N = W.shape[0]
    if W.shape != (N, K) or Y.shape != (B, N):
        raise ValueError("dense_fwd: shape mismatch")
    xp, ldx = _rows(X, "X")
    yp, ldy = _rows(Y, "Y")
    if pre is not None and pre.numel() != B * N:
        raise ValueError("dense_fwd: bad pre size")
    if stats is not None and stats.numel() != 2 * B:
        raise ValueError("dense_fwd: bad stats size")
    _lib.call("mmsa_dense_fwd", xp, ldx, _chk(W), _opt(bias), _opt(gamma), _opt(beta), yp, ldy, _opt(pre), _opt(stats),
              B, K, N, epilogue, float(drop_p), int(seed), _opt(step), _stream())


def dense_bwd(dY, X, pre, stats, W, gamma, beta, dPre, dLn, dX, dW, db, dgamma, dbeta, epilogue=EPI_NONE, drop_p=0.0,
              seed=0, accumulate_dx=False, step=None):
    B, K = X.shape
    N = W.shape[0]
    if dY.shape != (B, N) or dPre.numel() != B * N:
        raise ValueError("dense_bwd: shape mismatch")
    dyp, ldy = _rows(dY, "dY")
    xp, ldx = _rows(X, "X")
    dxp, lddx = (None, 0) if dX is None else _rows(dX, "dX")
    if dX is not None and dX.shape != (B, K):
        raise ValueError("dense_bwd: bad dX shape")
    if dW is not None and (dW.numel() != N * K or db is None or db.numel() != N):
        raise ValueError("dense_bwd: bad dW/db")
    _lib.call("mmsa_dense_bwd", dyp, ldy, xp, ldx, _opt(pre), _opt(stats), _chk(W), _opt(gamma), _opt(beta), _chk(dPre),
              _opt(dLn), dxp, lddx, _opt(dW), _opt(db), _opt(dgamma), _opt(dbeta), B, K, N, epilogue, float(drop_p),
              int(seed), int(accumulate_dx), _opt(step), _stream())


def dense_bwd_w(dPre, dLn, X, pre, stats, dW, db, dgamma, dbeta, epilogue):
    B, K = X.shape
    N = dW.shape[0]
    if dPre.numel() != B * N or dW.numel() != N * K or db.numel() != N:
        raise ValueError("dense_bwd_w: shape mismatch")
    xp, ldx = _rows(X, "X")
    _lib.call("mmsa_dense_bwd_w", _chk(dPre), _opt(dLn), _opt(pre), _opt(stats), xp, ldx, _chk(dW), _chk(db),
              _opt(dgamma), _opt(dbeta), B, K, N, epilogue, _stream())


def ln_scale_fwd(x, gamma, beta, scale, keep, y, stats):
    B, N = x.shape
    yp, ldy = _rows(y, "y")
    if y.shape != (B, N) or stats.numel() != 2 * B:
        raise ValueError("ln_scale_fwd: bad sizes")
    _lib.call("mmsa_ln_scale_fwd", _chk(x), _chk(gamma), _chk(beta), _chk(scale), _opt(keep), yp, ldy, _chk(stats), B, N,
              _stream())


def ln_scale_bwd(dy, x, gamma, beta, scale, keep, stats, dx, dgamma, dbeta, dscale, scratch):
    B, N = x.shape
    dyp, ldy = _rows(dy, "dy")
    if dy.shape != (B, N) or dx.numel() != B * N or scratch.numel() < B * N + B:
        raise ValueError("ln_scale_bwd: bad sizes")
    _lib.call("mmsa_ln_scale_bwd", dyp, ldy, _chk(x), _chk(gamma), _chk(beta), _chk(scale), _opt(keep), _chk(stats),
              _chk(dx), _chk(dgamma), _chk(dbeta), _chk(dscale), _chk(scratch), B, N, _stream())


HEAD_SLOTS = (
    "a_feat t_feat an_g an_b a_scale tn_g tn_b t_scale keep_a keep_t "
    "wpa bpa gpa bepa wpt bpt gpt bept watt batt w0 b0 g0 be0 w1 b1 g1 be1 w2 b2 "
    "an tn an_stats tn_stats pa_pre pa_stats pt_pre pt_stats P att wts fused "
    "c0_pre c0_stats h1 c1_pre c1_stats h2 logits "
    "dlogits c1_dpre c1_dln c0_dpre c0_dln att_dpre pa_dpre pa_dln pt_dpre pt_dln "
    "an_dln tn_dln a_dspart t_dspart d_afeat d_tfeat "
    "dwpa dbpa dgpa dbepa dwpt dbpt dgpt dbept dwatt dbatt dw0 db0 dg0 dbe0 dw1 db1 dg1 dbe1 dw2 db2 "
    "dan_g dan_b da_scale dtn_g dtn_b dt_scale labels step").split()


def head_both_table(named: dict):
    """ctypes pointer table for the fused head entry points from {slot name: tensor or None}."""
    n = _lib.query("mmsa_head_both_slots")
    if n != len(HEAD_SLOTS):
        raise _lib.MmsaError("fused head: slot table of the library and of the binding differ")
    unknown = set(named) - set(HEAD_SLOTS)
    if unknown:
        raise KeyError(f"fused head: unknown slots {sorted(unknown)}")
    return _parr([0 if named.get(k) is None else _chk(named[k], torch.int64 if k == "labels" else F32, name=k)
                  for k in HEAD_SLOTS])


def head_both(which: str, table, dims, drop_p=0.0, seed=0, grad_scale=1.0):
    """which: 'fwd' | 'bwd' | 'bwd_params'; dims = (B, A, T, Hd, N0, N1, C)."""
    if which == "bwd_params":
        _lib.call("mmsa_head_both_bwd_params", table, len(HEAD_SLOTS), *[int(d) for d in dims], _stream())
    elif which == "fwd":
        _lib.call("mmsa_head_both_fwd", table, len(HEAD_SLOTS), *[int(d) for d in dims], float(drop_p), int(seed),
                  float(grad_scale), _stream())
    else:
        _lib.call("mmsa_head_both_bwd", table, len(HEAD_SLOTS), *[int(d) for d in dims], float(drop_p), int(seed),
                  _stream())


def fusion_mix_fwd(P, att, wts, fused):
    B, H2 = P.shape
    if att.shape != (B, 2) or wts.numel() != 2 * B or fused.shape != (B, H2 // 2):
        raise ValueError("fusion_mix_fwd: bad sizes")
    _lib.call("mmsa_fusion_mix_fwd", _chk(P), _chk(att), _chk(wts), _chk(fused), B, H2 // 2, _stream())


def fusion_mix_bwd(dfused, P, wts, dP, datt):
    B, H2 = P.shape
    if dfused.shape != (B, H2 // 2) or dP.shape != (B, H2) or datt.numel() != 2 * B:
        raise ValueError("fusion_mix_bwd: bad sizes")
    _lib.call("mmsa_fusion_mix_bwd", _chk(dfused), _chk(P), _chk(wts), _chk(dP), _chk(datt), B, H2 // 2, _stream())


def pair_attn_mix_fwd(M, scores, wts, Wm):
    B, F2 = M.shape
    if scores.numel() != 2 * B or wts.numel() != 2 * B or Wm.shape != (B, F2):
        raise ValueError("pair_attn_mix_fwd: bad sizes")
    _lib.call("mmsa_pair_attn_mix_fwd", _chk(M), _chk(scores), _chk(wts), _chk(Wm), B, F2 // 2, _stream())


def pair_attn_mix_bwd(dWm, M, wts, dM, dscores, accumulate=False):
    B, F2 = M.shape
    if dWm.shape != (B, F2) or dM.shape != (B, F2) or dscores.numel() != 2 * B:
        raise ValueError("pair_attn_mix_bwd: bad sizes")
    _lib.call("mmsa_pair_attn_mix_bwd", _chk(dWm), _chk(M), _chk(wts), _chk(dM), _chk(dscores), B, F2 // 2,
              int(accumulate), _stream())


def gate_mix_fwd(Wm, gate, fused):
    B, F2 = Wm.shape
    if gate.shape != (B, F2 // 2) or fused.shape != (B, F2 // 2):
        raise ValueError("gate_mix_fwd: bad sizes")
    _lib.call("mmsa_gate_mix_fwd", _chk(Wm), _chk(gate), _chk(fused), B, F2 // 2, _stream())


def gate_mix_bwd(dfused, Wm, gate, dWm, dgate, accumulate=False):
    B, F2 = Wm.shape
    if dfused.shape != (B, F2 // 2) or (dWm is not None and dWm.shape != (B, F2)):
        raise ValueError("gate_mix_bwd: bad sizes")
    _lib.call("mmsa_gate_mix_bwd", _chk(dfused), _chk(Wm), _chk(gate), _opt(dWm), _opt(dgate), B, F2 // 2,
              int(accumulate), _stream())


def softmax_ce(logits, labels, probs, loss, dlogits):
    B, C = logits.shape
    if probs.shape != (B, C) or (dlogits is not None and dlogits.shape != (B, C)):
        raise ValueError("softmax_ce: bad sizes")
    if labels is not None and labels.numel() != B:
        raise ValueError("softmax_ce: bad labels size")
    _lib.call("mmsa_softmax_ce", _chk(logits), _opt(labels, torch.int64, "labels"), _chk(probs), _opt(loss),
              _opt(dlogits), B, C, _stream())


def bn1d_relu_fwd(x, gamma, beta, rmean, rvar, nbt, training, y, coeff, drop_p=0.0, seed=0, step=None):
    B, N = x.shape
    if y.shape != (B, N) or coeff.numel() != 4 * N:
        raise ValueError("bn1d_relu_fwd: bad sizes")
    _lib.call("mmsa_bn1d_relu_fwd", _chk(x), _chk(gamma), _chk(beta), _chk(rmean), _chk(rvar), _opt(nbt, torch.int64),
              BN_MOMENTUM, BN_EPS, int(training), _chk(y), _chk(coeff), B, N, float(drop_p), int(seed), _opt(step),
              _stream())


def bn1d_relu_bwd(dy, x, coeff, dx, dgamma, dbeta, training, drop_p=0.0, seed=0, step=None):
    B, N = x.shape
    if dy.shape != (B, N) or dx.shape != (B, N):
        raise ValueError("bn1d_relu_bwd: bad sizes")
    _lib.call("mmsa_bn1d_relu_bwd", _chk(dy), _chk(x), _chk(coeff), _chk(dx), _chk(dgamma), _chk(dbeta), B, N,
              int(training), float(drop_p), int(seed), _opt(step), _stream())


def dropout(x, y, drop_p, seed, step=None):
    if x.numel() != y.numel():
        raise ValueError("dropout: size mismatch")
    _lib.call("mmsa_dropout", _chk(x), _chk(y), x.numel(), float(drop_p), int(seed), _opt(step), _stream())


# ---------------------------------------------------------------------------------------------- optimiser
def param_norms(flat, table, tnorm, sum_out, chunk_sums_ready=False):
    """Per-tensor norms of the flat parameter buffer.  With chunk_sums_ready the per-chunk sums of squares left in
    ``table.chunk_ss_p`` by the previous AdamW step are used and the 38 MB pass over the parameters is skipped."""
    _lib.call("mmsa_param_norms", _chk(flat), _chk(table.chunk_off, torch.int64), _chk(table.chunk_len, torch.int32),
              _chk(table.chunk_tensor, torch.int32), table.nchunks, _chk(table.tensor_first_chunk, torch.int32),
              table.ntensors, _chk(table.chunk_ss_p, torch.float64), _chk(tnorm), _opt(sum_out),
              int(chunk_sums_ready), _stream())


def check_labels(labels, num_classes: int, err):
    """Raises bit 1 of the sticky flag ``err`` (int32[1]) if a label lies outside [0, num_classes)."""
    _lib.call("mmsa_check_labels", _chk(labels, torch.int64, "labels"), labels.numel(), int(num_classes),
              _chk(err, torch.int32, "err"), _stream())


def l2_grad_add(gflat, pflat, tnorm, lam, table, scale=None):
    """gflat += lam * scale * p / ||p|| per tensor; ``scale``: optional one-element device tensor."""
    _lib.call("mmsa_l2_grad_add", _chk(gflat), _chk(pflat), _chk(tnorm), float(lam), _opt(scale),
              _chk(table.chunk_off, torch.int64),
              _chk(table.chunk_len, torch.int32), _chk(table.chunk_tensor, torch.int32), table.nchunks, _stream())


def grad_norm(gflat, table, state, pflat=None, tnorm=None, lam=0.0):
    """state[4] <- global norm of gflat (+ lam * p / ||p|| per tensor when pflat / tnorm are given)."""
    _lib.call("mmsa_grad_norm", _chk(gflat), _opt(pflat), _opt(tnorm), float(lam), _chk(table.chunk_off, torch.int64),
              _chk(table.chunk_len, torch.int32), _chk(table.chunk_tensor, torch.int32), table.nchunks,
              _chk(table.tensor_first_chunk, torch.int32), table.ntensors, _chk(table.chunk_ss, torch.float64),
              _chk(state), _stream())


def grad_norm_finish(gflat, table, state, beta1, beta2, max_norm, do_clip, pflat=None, tnorm=None, lam=0.0,
                     sparse=None):
    """:func:`grad_norm` that also derives the step's scalars (step count, bias corrections, clip coefficient) in its
    second launch; :func:`adamw_step` then takes ``scalars_ready=True``.  ``sparse = (tensor index, row width, ids,
    first)``: that tensor is a table whose gradient is zero outside the rows ``ids[q]`` with ``first[q] > 0`` (the
    ``seg_cnt`` part of :func:`token_groups`' result); its share of the norm is read from those rows only."""
    sp = (-1, 0, 0, 0, 0, None, None)
    if sparse is not None:
        t, width, ids, first = sparse
        n = ids.numel()
        if first.numel() != n or not 0 <= t < table.ntensors:
            raise ValueError("grad_norm_finish: bad row-sparse tensor description")
        sp = (t, table.first_host[t], table.first_host[t + 1] - table.first_host[t], int(width), n,
              _chk(ids, torch.int64, "ids"), _chk(first, torch.int32, "first"))
    _lib.call("mmsa_grad_norm_finish", _chk(gflat), _opt(pflat), _opt(tnorm), float(lam),
              _chk(table.chunk_off, torch.int64), _chk(table.chunk_len, torch.int32),
              _chk(table.chunk_tensor, torch.int32), table.nchunks,
              _chk(table.tensor_first_chunk, torch.int32), table.ntensors, _chk(table.chunk_ss, torch.float64),
              _chk(state), float(beta1), float(beta2), float(max_norm), int(do_clip), *sp, _stream())


def adamw_step(p, g, m, v, table, state, lr, beta1, beta2, eps, wd, max_norm, do_clip, tnorm=None, lam=0.0,
               emit_param_sums=False, scalars_ready=False):
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or state.numel() < 8:
        raise ValueError("adamw_step: size mismatch")
    _lib.call("mmsa_adamw_step", _chk(p), _chk(g), _chk(m), _chk(v), _chk(table.chunk_off, torch.int64),
              _chk(table.chunk_len, torch.int32), _chk(table.chunk_tensor, torch.int32), table.nchunks, _opt(tnorm),
              float(lam), _chk(state), float(lr), float(beta1), float(beta2), float(eps), float(wd), float(max_norm),
              int(do_clip), _chk(table.chunk_ss_p, torch.float64) if emit_param_sums else None, int(scalars_ready),
              _stream())


# Diagnostic time stamps (tools/step_stamps.py): STAMPS = (int64 device buffer, {label: slot}) switches them on.
STAMPS = None


def stamp(label: str):
    """Records the GPU's real-time counter when the current stream reaches this point (no-op unless STAMPS is set)."""
    if STAMPS is None:
        return
    buf, slots = STAMPS
    slot = slots.setdefault(label, len(slots))
    if slot >= buf.numel():
        raise ValueError("stamp buffer too small")
    _lib.call("mmsa_stamp", _chk(buf, torch.int64), slot, _stream())


def scale_by_state(g, state, idx):
    _lib.call("mmsa_scale_by_state", _chk(g), g.numel(), _chk(state), idx, _stream())


def scalar_axpy(dst, a, b, alpha):
    _lib.call("mmsa_scalar_axpy", _chk(dst), _chk(a), _chk(b), float(alpha), _stream())

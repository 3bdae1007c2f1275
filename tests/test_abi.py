"""CPU tests of the drop-in boundary: the shared library loads and exports every symbol include/mmsa.h declares,
argument checks reject bad shapes without launching anything, and host-only helpers work without a GPU."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from mmsa_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m mmsa_amd.build"
    protos = _lib.parse_header(os.path.join(ROOT, "include", "mmsa.h"))
    assert len(protos) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing
    _lib.load()
    # nothing but the declared C ABI is exported with the mmsa_ prefix
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line and "mmsa_" in line}
    assert exported == set(protos), exported ^ set(protos)


def test_host_table_helper_and_geometry_queries():
    from mmsa_amd import _lib
    win = np.zeros(1024, np.float32)
    t5 = np.zeros(512, np.float32)
    t10 = np.zeros(1026, np.float32)
    _lib.call("mmsa_logmel_tables_host", win.ctypes.data_as(ctypes.c_void_p), t5.ctypes.data_as(ctypes.c_void_p),
              t10.ctypes.data_as(ctypes.c_void_p))
    n = np.arange(1024)
    np.testing.assert_allclose(win, 0.5 - 0.5 * np.cos(2 * np.pi * n / 1024), atol=1e-7)
    assert t5[0] == 1.0 and abs(t5[2 * 128] - 0.0) < 1e-7 and abs(t5[2 * 128 + 1] + 1.0) < 1e-7
    assert abs(t10[2 * 512] + 1.0) < 1e-7
    assert _lib.query("mmsa_conv_tile_rows", 32, 32, 65, 64) in (64, 128)
    assert _lib.query("mmsa_conv_wgrad_workspace_floats", 32, 32, 65, 64, 64, 9, None) > 0
    assert _lib.query("mmsa_resblock_bwd_scratch_floats", 32, 32, 65, 64) > 0
    with pytest.raises(_lib.MmsaError):
        _lib.query("mmsa_resblock_bwd_scratch_floats", 32, 32, 65, 48)      # channels must be 64/128/256


def test_resample_filter_bank_matches_oracle_and_has_unit_dc_gain():
    """Host-built filter bank of the resampler == the oracle's float64 construction, for the rates the datasets use."""
    from mmsa_amd import _lib
    from oracle import resample_ref
    for sr, want in ((16000, (320, 441, 7, 334)), (44100, (2, 1, 13, 28)), (48000, (320, 147, 14, 348)),
                     (8000, (160, 441, 7, 174))):
        geo = (ctypes.c_int * 4)()
        ptrs = [ctypes.cast(ctypes.byref(geo, 4 * i), ctypes.c_void_p) for i in range(4)]
        _lib.call("mmsa_resample_filters_host", sr, 22050, 6, 0.99, None, *ptrs)
        orig, new, width, K = (int(v) for v in geo)
        assert (orig, new, width, K) == want
        ld = (new + 3) // 4 * 4                                  # rows are padded to 16-byte multiples
        kern_t = np.ones((K, ld), np.float32)
        _lib.call("mmsa_resample_filters_host", sr, 22050, 6, 0.99, kern_t.ctypes.data_as(ctypes.c_void_p), *ptrs)
        filt, w, o, n = resample_ref.sinc_filters(sr, 22050)
        assert (o, n, w) == (orig, new, width) and filt.shape == (new, K)
        np.testing.assert_allclose(kern_t[:, :new].T, filt, rtol=0, atol=2e-7)
        assert not kern_t[:, new:].any()
        # every output phase passes DC at the gain of a windowed sinc cut just below Nyquist
        assert np.abs(filt.sum(1) - 1.0).max() < 2e-3
    lib = _lib.load()
    assert lib.mmsa_resample_fwd(None, 1, 100, 100, None, 320, 441, 7, 334, None, 66150, 1, None) == -1


def test_resample_oracle_reconstructs_a_tone():
    """The restated sinc resampler maps a 440 Hz tone at 16 kHz onto the same tone at 22.05 kHz (|err| < 1e-3 away
    from the edges), yields ceil(new * T / orig) samples, and `prepare` pads / trims to 66 150."""
    from oracle import resample_ref
    t = np.arange(16000) / 16000.0
    x = np.sin(2 * np.pi * 440 * t).astype(np.float32)
    y = resample_ref.resample(x, 16000, 22050)
    assert y.shape == (22050,)
    t2 = np.arange(22050) / 22050.0
    assert np.abs(y[200:-200] - np.sin(2 * np.pi * 440 * t2)[200:-200]).max() < 1e-3
    assert resample_ref.resample(x[:1001], 16000, 22050).shape == (1380,)      # ceil(441 * 1001 / 320)
    stereo = np.stack([x, -x * 0.5])
    out = resample_ref.prepare(stereo, 16000)
    assert out.shape == (66150,) and np.all(out[22050:] == 0)
    np.testing.assert_allclose(out[:22050], 0.25 * y, atol=1e-6)
    long = resample_ref.prepare(np.tile(x, 4)[None], 16000)
    assert long.shape == (66150,) and np.abs(long[-100:]).max() > 0.1


def test_bad_arguments_are_rejected_before_any_launch():
    from mmsa_amd import _lib
    lib = _lib.load()
    # null pointers / bad geometry -> MMSA_ERR_ARG (-1), nothing touches the (absent) GPU
    assert lib.mmsa_conv_fwd(None, None, None, None, 1, 1, 1, 32, 64, 9, None, None, 0, None) == -1
    assert lib.mmsa_embedding_gather(None, None, None, 4, 256, 1000, None, None) == -1
    assert lib.mmsa_adamw_step(None, None, None, None, None, None, None, 4, None, 0.0, None, 1e-3, 0.9, 0.999, 1e-8,
                               0.0, 1.0, 1, None, 0, None) == -1
    with pytest.raises(_lib.MmsaError):
        _lib.call("mmsa_softmax_ce", None, None, None, None, None, 4, 3, None)


def test_mel_filterbank_shape_and_support():
    from mmsa_amd.frontend import htk_mel_filterbank
    from oracle import frontend_ref
    fb = htk_mel_filterbank()
    assert fb.shape == (513, 64)
    assert int((fb > 0).sum()) == 998                       # non-zeros of the 64 HTK triangles at n_fft 1024
    assert (fb.sum(0) > 0).all()                            # no empty filter
    assert np.array_equal(fb.numpy(), frontend_ref.mel_filterbank().numpy())


def test_synthetic_data_is_deterministic_and_well_formed():
    from mmsa_amd import data
    a, b = data.make_batch(7, 5, with_wave=True), data.make_batch(7, 5, with_wave=True)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a["audio"].shape == (5, 1, 64, 130) and a["wave"].shape == (5, 66150)
    ids = a["input_ids"]
    assert ids.dtype == np.int64 and (ids[:, 0] == 101).all() and ((ids == 102).sum(1) == 1).all()
    assert np.array_equal(a["attention_mask"], (ids != 0).astype(np.int64))
    assert set(np.unique(data.label_batch(0, 200))) == {0, 1, 2}
    rows = data.csv_rows(0, 3)
    assert rows[0] == "Audio_ID;Sentiment;Utterance" and rows[1].split(";")[1] in data.SENTIMENTS


def test_noise_gate_plan_constants_and_oracle_behaviour():
    """Host side of the noise reduction (preprocess.py:142-148): the frame plan of the C ABI agrees with scipy's STFT
    framing of the padded chunk, the gate constants agree with the oracle's, argument checks reject bad calls, and the
    oracle itself behaves like a noise gate (a stationary floor is suppressed, a burst passes, silence gives NaN as
    0/0 does in the package)."""
    import torch
    from scipy.signal import stft
    from mmsa_amd import _lib, ops
    from oracle import denoise_ref
    lib = _lib.load()
    for L, pad in ((66150, 30000), (1, 30000), (3000, 0), (600000, 30000), (16000, 5000)):
        geo = (ctypes.c_int * 3)()
        need = ctypes.c_long()
        _lib.call("mmsa_denoise_plan", L, pad, *[ctypes.cast(ctypes.byref(geo, 4 * i), ctypes.c_void_p) for i in range(3)],
                  ctypes.cast(ctypes.byref(need), ctypes.c_void_p))
        nseg, f_lo, nfr = (int(v) for v in geo)
        if L + 2 * pad <= 200000:
            _, _, z = stft(np.zeros(L + 2 * pad), nfft=1024, noverlap=768, nperseg=1024, padded=False)
            assert z.shape == (513, nseg)
        # frames f_lo .. f_lo+nfr-1 are exactly those whose 1024 samples (centred on 256 f) touch [pad, pad + L)
        touching = [f for f in range(nseg) if 256 * f + 512 > pad and 256 * f - 512 < pad + L]
        assert touching == list(range(f_lo, f_lo + nfr))
        assert need.value == nseg * 516 * 3 + nfr * 1024
    assert lib.mmsa_denoise_plan(0, 30000, None, None, None, None) == -1
    geo4 = [ctypes.cast(ctypes.byref((ctypes.c_long * 1)()), ctypes.c_void_p) for _ in range(4)]
    assert lib.mmsa_denoise_plan(777, 0, *geo4) == -1        # chunk shorter than one STFT frame
    null = [None] * 4                                  # window, tw512, tw1024, vt
    assert lib.mmsa_denoise_fwd(None, 100, 100, 0, 100, 30000, 1, None, *null, 9, None, 23, 0.005, 2.0, 10.0, 1.0,
                                None, None, 100, None) == -1
    filt = ops.denoise_filters(22050, "cpu")
    vf, vt = denoise_ref.smoothing_taps(22050)
    assert filt["vf"].numel() == 23 and filt["vt"].numel() == 9
    np.testing.assert_allclose(filt["vf"].numpy(), vf / vf.sum(), rtol=1e-6)
    np.testing.assert_allclose(filt["vt"].numpy(), vt / vt.sum(), rtol=1e-6)
    assert abs(filt["b"] - denoise_ref.iir_coefficient(22050)) < 1e-15
    with pytest.raises(ValueError):
        ops.denoise_filters(400000, "cpu")            # smoothing narrower than one bin
    with pytest.raises(_lib.MmsaError):
        ops.denoise(torch.zeros(1, 100), filt, {})    # host tensors are refused: there is no CPU path

    g = np.random.default_rng(0)
    n = 44100
    t = np.arange(n) / 22050.0
    burst = 0.3 * np.sin(2 * np.pi * 440 * t) * ((t > 0.9) & (t < 1.1))
    y = (burst + 0.02 * g.standard_normal(n)).astype(np.float32)
    out = denoise_ref.reduce_noise(y, 22050)
    assert out.dtype == np.float32 and out.shape == y.shape
    assert out[:15000].std() < 0.5 * y[:15000].std()                    # the floor drops
    assert out[20500:23500].std() > 0.5 * y[20500:23500].std()          # the onset of the burst passes
    assert np.isnan(denoise_ref.reduce_noise(np.zeros(3000, np.float32), 22050)).all()
    # chunked processing only differs from whole-signal processing through the shorter context
    whole = denoise_ref.reduce_noise(y, 22050)
    parts = denoise_ref.reduce_noise(y, 22050, chunk_size=20000, padding=15000)
    assert np.isfinite(parts).all() and np.abs(whole - parts).max() < 0.3 * np.abs(whole).max()


def test_mel_filterbank_against_an_independent_htk_implementation():
    """transformers.audio_utils.mel_filter_bank(norm=None, mel_scale="htk") is a second, independently written HTK
    filterbank (float64 numpy).  It pins nothing -- the reference calls torchaudio -- but two implementations of the
    same published formula must agree to float32 rounding of the triangle edges."""
    au = pytest.importorskip("transformers.audio_utils")
    from oracle import frontend_ref
    other = au.mel_filter_bank(num_frequency_bins=513, num_mel_filters=64, min_frequency=0.0, max_frequency=11025.0,
                               sampling_rate=22050, norm=None, mel_scale="htk")
    ours = frontend_ref.mel_filterbank().numpy()
    assert other.shape == ours.shape == (513, 64)
    assert np.abs(other - ours).max() < 2e-5
    ours64 = frontend_ref.mel_filterbank(dtype=__import__("torch").float64).numpy()
    assert np.abs(other - ours64).max() < 1e-9
    # same support except where a triangle's corner falls within rounding distance of an FFT bin
    assert int(((other > 0) != (ours > 0)).sum()) <= 2


def test_zero_time_shift_raises_and_stops_drawing(monkeypatch):
    """preprocess.py:65,70: `audio_shift[:, 0:] = audio[:, :-0]` raises for a zero shift; __call__ (`:172-174`) then
    returns zeros without consuming further random numbers.  The host-side path must do the same."""
    import random
    import torch
    from mmsa_amd.frontend import AudioTransform
    seq = iter([0.9, 0.0, 0.7, 0.9, 0.9, 0.9])     # shift branch taken, shift fraction 0 -> zero samples, direction
    calls = []

    def fake():
        calls.append(1)
        return next(seq)
    monkeypatch.setattr(random, "random", fake)
    t = AudioTransform(augment=True, device="cpu", waveform_only=True, noise_reduce=False)
    wave = torch.randn(1, 30000)
    out = t(wave, 22050)
    assert out.shape == (1, 66150) and float(out.abs().max()) == 0.0
    assert len(calls) == 3                          # no draw for the noise or the index-resampling decision
    seq = iter([0.9, 0.0, 0.7])
    with pytest.raises(RuntimeError):
        AudioTransform(augment=True, device="cpu", waveform_only=True, noise_reduce=False, strict=True)(wave, 22050)


def test_torch_custom_ops_are_registered_for_the_gpu_only():
    """torch.ops.mmsa.* exist after importing the package (the thin custom-op layer over the C ABI); they have no
    CPU kernel, so a CPU call is refused by the dispatcher instead of silently computing somewhere else."""
    import torch
    import mmsa_amd  # noqa: F401
    from mmsa_amd import torch_ops
    assert len(set(torch_ops.REGISTERED)) == len(torch_ops.REGISTERED) >= 25
    for name in torch_ops.REGISTERED:
        op = getattr(torch.ops.mmsa, name)
        assert "mmsa::" + name in str(op.default._schema)
    with pytest.raises(NotImplementedError):
        torch.ops.mmsa.bn_relu_apply(torch.zeros(8, 4), torch.zeros(16), torch.zeros(8, 4), 8, 4)


def test_bench_reads_the_committed_rocprof_summary():
    """bench.py names its dominant kernel after the newest committed rocprofv3 summary: the file must parse, hold the
    matrix kernels under the names the launch wrappers use, and the traffic table must cover the dominant one."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench._canon('"void conv_igemm_kernel<64, 64, 2, 2, true, false, 5>"') == \
        "conv_igemm_kernel<64,64,2,2,true,false,5>"
    rows, src = bench.rocprof_summary()
    assert src and src.startswith("profiles/r") and src.endswith("_kernel_stats.csv")
    matrix = [k for k in rows if k.startswith(("conv_igemm_kernel<", "conv_igemm3_kernel<", "conv_wgrad2_kernel<"))]
    assert len(matrix) >= 8
    top = max(matrix, key=lambda k: rows[k][1])
    avg_us, total_us, calls = rows[top]
    assert calls > 0 and abs(avg_us * calls - total_us) < 0.01 * total_us
    tag = src.split("/")[1].split("_")[0]
    traffic = json.load(open(os.path.join(root, "profiles", tag + "_pmc_traffic.json")))
    assert top in {bench._canon(k) for k in traffic}

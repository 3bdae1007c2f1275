"""GPU parity tests, kernel by kernel, through the C ABI (mmsa_amd.ops -> libmmsa_hip.so).

Checker: torch fp32/fp64 CPU arithmetic and the oracle under oracle/.  Tolerances are stated per test;
index/byte work (gather, argmax) must be bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import batch_tensors

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mmsa_amd import ops
    return ops


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, tol):
    e = rel_err(a, b)
    assert e < tol, f"relative error {e:.3e} exceeds {tol:.1e}"


# ------------------------------------------------------------------------------------------------ front end
def test_logmel_matches_oracle():
    from mmsa_amd import data
    from mmsa_amd.frontend import LogMelFrontEnd
    from oracle import frontend_ref
    wave = torch.from_numpy(data.waveform_batch(0, 6))
    fe = LogMelFrontEnd(DEV)
    got = fe(wave.to(DEV)).cpu()
    assert got.shape == (6, 1, 64, 130)
    ref64 = torch.from_numpy(frontend_ref.logmel_f64(wave.numpy()))
    ref32 = frontend_ref.logmel_f32(wave)
    # mel features within 1e-4 relative (of the map's dynamic range; values are z-scores clamped to +-6)
    check(got, ref64, 1e-4)
    check(got, ref32, 2e-4)
    assert abs(float(got.mean())) < 1e-3 and abs(float(got.std()) - 1.0) < 0.05


def test_logmel_edge_cases():
    """Short input is zero-padded, long input trimmed, silence gives zeros (std 0 -> 0/1e-8), strided rows work."""
    from mmsa_amd.frontend import LogMelFrontEnd
    from oracle import frontend_ref
    fe = LogMelFrontEnd(DEV)
    g = torch.Generator().manual_seed(3)
    short = torch.randn(2, 30000, generator=g) * 0.1
    long = torch.randn(2, 90000, generator=g) * 0.1
    for w in (short, long):
        got = fe(w.to(DEV)).cpu()
        ref = torch.from_numpy(frontend_ref.logmel_f64(w.numpy()))
        check(got, ref, 1e-4)
    sil = fe(torch.zeros(1, 66150, device=DEV)).cpu()
    assert torch.equal(sil, torch.zeros_like(sil))
    base = torch.randn(3, 70000, generator=g) * 0.05
    view = base.to(DEV)[:, :66150]          # row stride 70000
    got = fe(view).cpu()
    ref = torch.from_numpy(frontend_ref.logmel_f64(base[:, :66150].numpy()))
    check(got, ref, 1e-4)


@pytest.mark.parametrize("sr,C,T,B", [(16000, 1, 48000, 3), (16000, 2, 20001, 2), (8000, 1, 30000, 2),
                                      (44100, 2, 150000, 2), (48000, 1, 100000, 1)])
def test_resample_matches_oracle(sr, C, T, B):
    """Mono mix + sinc resampling + pad/trim vs the oracle accumulated in float64: |err| < 2e-6 * max|y| (fp32
    accumulation over <= 348 taps); the zero padding past the resampled length is exact."""
    from mmsa_amd.frontend import Resampler
    from oracle import resample_ref
    g = torch.Generator().manual_seed(sr + T)
    pcm = torch.randn(B, C, T, generator=g) * 0.2
    rs = Resampler(sr, 22050, DEV)
    got = rs(pcm.to(DEV)).cpu()
    assert got.shape == (B, 66150)
    ref = np.stack([resample_ref.prepare(pcm[b].numpy(), sr, dtype=np.float64) for b in range(B)])
    check(got, torch.from_numpy(ref), 2e-6)
    n_res = rs.resampled_length(T)
    if n_res < 66150:
        assert torch.equal(got[:, n_res:], torch.zeros(B, 66150 - n_res))
    full = rs(pcm.to(DEV), n_out=None).cpu()
    assert full.shape == (B, n_res)
    n = min(n_res, 66150)
    assert torch.equal(full[:, :n], got[:, :n])


def test_audio_transform_accepts_16k_input():
    """The reference's per-utterance call with sr != 22050 (preprocess.py:125-134): resample, then the mel stage."""
    from mmsa_amd.frontend import AudioTransform
    from oracle import frontend_ref, resample_ref
    g = torch.Generator().manual_seed(11)
    audio = torch.randn(2, 40000, generator=g) * 0.1
    spec = AudioTransform(augment=False, device=DEV, noise_reduce=False)(audio, 16000)
    assert spec.shape == (1, 64, 130) and not spec.is_cuda
    wave = resample_ref.prepare(audio.numpy(), 16000, dtype=np.float64)
    check(spec, torch.from_numpy(frontend_ref.logmel_f64(wave[None]))[0], 1e-4)


# ------------------------------------------------------------------------------------------------ noise reduction
def _noisy_speechlike(B, n, seed):
    """Tone bursts with moving pitch over a noise floor, so that the gate opens and closes within each row."""
    g = np.random.default_rng(seed)
    t = np.arange(n) / 22050.0
    rows = []
    for b in range(B):
        f0, on, off = g.uniform(150, 900), g.uniform(0.1, 0.4) * n, g.uniform(0.6, 0.9) * n
        env = ((np.arange(n) > on) & (np.arange(n) < off)).astype(np.float64)
        sig = 0.3 * env * np.sin(2 * np.pi * f0 * t * (1 + 0.2 * t)) + 0.1 * env * np.sin(2 * np.pi * 3.1 * f0 * t)
        rows.append(sig + g.uniform(0.005, 0.03) * g.standard_normal(n))
    return np.stack(rows).astype(np.float32)


def test_noise_reduction_matches_oracle():
    """Spectral gate vs the float64 restatement built on scipy's stft / filtfilt / fftconvolve / istft: the fp32
    transforms with double-precision recurrences stay within 5e-6 of the row's peak (measured 3e-7).  The noise floor
    ahead of the burst must drop by more than half."""
    from mmsa_amd.frontend import NoiseReducer
    from oracle import denoise_ref
    B, n = 3, 52000
    y = _noisy_speechlike(B, n, 3)
    nr = NoiseReducer(22050, DEV)
    got = nr(torch.from_numpy(y).to(DEV)).cpu()
    assert got.shape == (B, n)
    for b in range(B):
        ref = denoise_ref.reduce_noise(y[b].astype(np.float64), 22050)
        check(got[b], torch.from_numpy(ref), 5e-6)
    assert float(got[:, :3000].std()) < 0.5 * float(y[:, :3000].std())
    # deterministic: a second call gives the same bits
    assert torch.equal(got, nr(torch.from_numpy(y).to(DEV)).cpu())


def test_noise_reduction_rows_of_different_lengths_and_chunks():
    """Rows with their own valid lengths are gated as signals of those lengths (equal to separate calls, zeros behind
    them); long rows go chunk by chunk with the neighbouring samples as context, as the oracle does."""
    from mmsa_amd import ops
    from mmsa_amd.frontend import NoiseReducer
    from oracle import denoise_ref
    y = _noisy_speechlike(3, 40000, 5)
    lens = [40000, 23117, 777]
    nr = NoiseReducer(22050, DEV)
    x = torch.from_numpy(y).to(DEV)
    got = nr(x, lens=lens).cpu()
    for b, n in enumerate(lens):
        alone = nr(x[b:b + 1, :n].contiguous()).cpu()[0]
        assert torch.equal(got[b, :n], alone)
        assert not got[b, n:].any()
        check(alone, torch.from_numpy(denoise_ref.reduce_noise(y[b, :n].astype(np.float64), 22050)), 5e-6)
    for bad in ([40001, 5, 5], [0, 5, 5], [5, 5]):    # lengths are checked on the host before any launch
        with pytest.raises(ValueError):
            nr(x, lens=bad)
    # chunked: 3 chunks of 16000 with 5000 samples of context
    ch = ops.denoise(x, nr.filters, nr.tables, chunk=16000, pad=5000).cpu()
    for b in range(3):
        ref = denoise_ref.reduce_noise(y[b].astype(np.float64), 22050, chunk_size=16000, padding=5000)
        check(ch[b], torch.from_numpy(ref), 5e-6)


def test_audio_transform_applies_the_noise_gate_by_default():
    """preprocess.py:125-168 in full for a 16 kHz stereo utterance: mono, resample, noise gate on the whole resampled
    utterance, pad / trim, mel stage -- against the oracle chain."""
    from mmsa_amd.frontend import AudioTransform
    from oracle import denoise_ref, frontend_ref, resample_ref
    y = _noisy_speechlike(2, 52000, 8)                 # [C, T] at 16 kHz: 3.25 s, trimmed to 3 s after the gate
    tf = AudioTransform(device=DEV, strict=True)
    spec = tf(torch.from_numpy(y), 16000)
    res = resample_ref.resample(y.mean(axis=0, dtype=np.float32), 16000, 22050, np.float64).astype(np.float32)
    den = denoise_ref.reduce_noise(res, 22050)
    want = frontend_ref.logmel_f64(den[None, :66150])
    check(spec, torch.from_numpy(want)[0], 5e-4)          # z-scored dB map, relative to its range
    off = AudioTransform(device=DEV, strict=True, noise_reduce=False)(torch.from_numpy(y), 16000)
    assert float((off - spec).abs().max()) > 0.1


@pytest.mark.parametrize("C,n,sr", [(2, 154350, 44100), (1, 16000, 8000), (2, 44100, 11025), (1, 66151, 22050),
                                    (3, 3200, 16000), (1, 160000, 32000)])
def test_audio_transform_at_other_rates_channel_counts_and_lengths(C, n, sr):
    """The per-utterance call of preprocess.py:125-168 on what a corpus other than the reference's may hold: other
    sample rates (up- and down-sampling), one to three channels, 0.2 s to 5 s -- against the oracle chain; and the same
    utterance handed over as a 1-D, float64, device-resident or strided tensor gives the same bits."""
    from mmsa_amd.frontend import AudioTransform
    from oracle import denoise_ref, frontend_ref, resample_ref
    y = _noisy_speechlike(C, n, 7)
    mono = y.mean(axis=0, dtype=np.float32)
    res = resample_ref.resample(mono, sr, 22050, np.float64).astype(np.float32) if sr != 22050 else mono
    den = denoise_ref.reduce_noise(res, 22050)
    w = np.zeros(66150, np.float32)
    k = min(66150, len(den))
    w[:k] = den[:k]
    tf = AudioTransform(device=DEV, strict=True)
    spec = tf(torch.from_numpy(y), sr)
    assert spec.shape == (1, 64, 130)
    check(spec, torch.from_numpy(frontend_ref.logmel_f64(w[None]))[0], 5e-4)
    same = [torch.from_numpy(y).double(), torch.from_numpy(y).to(DEV), torch.from_numpy(np.repeat(y, 2, axis=1))[:, ::2]]
    if C == 1:
        same.append(torch.from_numpy(y[0]))
    for other in same:
        assert torch.equal(tf(other, sr), spec)


@pytest.mark.parametrize("augment", [False, True])
@pytest.mark.parametrize("noise_reduce", [False, True])
def test_batch_front_end_equals_the_per_utterance_transform(augment, noise_reduce):
    """The split transform -- CPU half in loader workers (mono mix), GPU half on the collated, zero-padded batch
    (resampling, augmentations, noise gate, pad / trim, mel stage; rows of different lengths, each handled at its own
    length) -- gives, utterance for utterance, what the per-utterance call of preprocess.py:125-168 gives; with
    augmentation the same `random` draws in the same order and the same noise seed."""
    import random
    from mmsa_amd.frontend import AudioTransform, collate_utterances
    lengths = [52000, 16000, 33333, 4000]
    utts = [torch.from_numpy(_noisy_speechlike(2, n, 20 + i)) for i, n in enumerate(lengths)]     # stereo, 16 kHz
    tf = AudioTransform(augment=augment, device=DEV, strict=True, noise_reduce=noise_reduce)
    batch = collate_utterances([tf.for_workers()(u, 16000) for u in utts])
    assert batch["pcm"].shape == (4, 52000)
    random.seed(5)
    torch.manual_seed(9)
    got = tf.batch_front_end()(batch)
    assert got.shape == (4, 1, 64, 130) and got.is_cuda
    if augment:
        # the batch path draws every utterance's decisions first and then one noise seed for the batch
        random.seed(5)
        drawn = [tf.draw_augmentation(-(-441 * n // 320)) for n in lengths]
        torch.manual_seed(9)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        for b, u in enumerate(utts):
            mono = u.mean(dim=0, keepdim=True)
            # utterance b alone, with its own decisions; the noise generator is keyed by (seed, row, sample), so the
            # utterance sits in row b of a batch whose other rows are silence
            pcm = torch.zeros(4, u.shape[1])
            pcm[b] = mono[0]
            alone = tf.process_batch(pcm, [1] * b + [u.shape[1]] + [1] * (3 - b), 16000,
                                     rows=[[0.0] * 4] * b + [drawn[b][0]] + [[0.0] * 4] * (3 - b),
                                     masks=[[0] * 4] * b + [drawn[b][1]] + [[0] * 4] * (3 - b), seed=seed)
            check(got[b], alone[b], 2e-5)
    else:
        for b, u in enumerate(utts):
            check(got[b], tf(u, 16000), 2e-5)
    # silence behind a short utterance, not the resampler's ringing: the padded batch row equals the lone utterance
    wave = tf.batch_front_end()(batch, return_wave=True) if not augment else None
    if wave is not None and not noise_reduce:
        n3 = -(-441 * 4000 // 320)
        assert not wave[3, n3:].any() and wave[3, :n3].any()


def test_batch_front_end_with_utterances_at_different_sample_rates():
    """A collated batch may hold files of several rates (multimodal_dataset.py:146 takes whatever rate the file has):
    the GPU half runs one launch sequence per rate and puts the maps back in batch order; with augmentation the
    decisions are still drawn utterance by utterance in batch order."""
    import random
    from mmsa_amd.frontend import AudioTransform, collate_utterances
    rates = [16000, 22050, 8000, 16000, 22050]
    utts = [torch.from_numpy(_noisy_speechlike(1, 20000 + 3000 * i, 40 + i)) for i in range(5)]
    tf = AudioTransform(augment=False, device=DEV, strict=True)
    batch = collate_utterances([tf.for_workers()(u, r) for u, r in zip(utts, rates)])
    assert batch["sr"].tolist() == rates
    got = tf.batch_front_end()(batch)
    assert got.shape == (5, 1, 64, 130)
    for b, (u, r) in enumerate(zip(utts, rates)):
        check(got[b], tf(u, r), 2e-5)
    ta = AudioTransform(augment=True, device=DEV, strict=True, noise_reduce=False)
    random.seed(3)
    torch.manual_seed(4)
    aug = ta.batch_front_end()(batch)
    random.seed(3)
    res_len = [ta._resampler(r).resampled_length(u.shape[1]) if r != 22050 else u.shape[1] for u, r in zip(utts, rates)]
    drawn = [ta.draw_augmentation(n) for n in res_len]          # batch order, whatever the rate
    # utterances that drew no waveform change and no mask must equal the plain transform's map
    plain = [b for b, (row, mask) in enumerate(drawn) if row == [0.0, 0.0, 0.0, 0.0] and mask == [0, 0, 0, 0]]
    changed = [b for b in range(5) if b not in plain]
    assert changed, "the seed should augment at least one utterance"
    for b in plain:
        check(aug[b], got[b], 2e-5)
    for b in changed:
        assert float((aug[b] - got[b]).abs().max()) > 1e-3 or drawn[b][0][0] != 0.0


# ------------------------------------------------------------------------------------------------ gather
def test_embedding_gather_bit_exact():
    ops = _ops()
    g = torch.Generator().manual_seed(0)
    table = torch.randn(30522, 256, generator=g)
    ids = torch.randint(0, 30522, (5, 128), generator=g)
    ids[0, :4] = torch.tensor([0, 101, 102, 30521])
    out = torch.empty(5, 128, 256, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embedding_gather(ids.to(DEV), table.to(DEV), out, err)
    assert torch.equal(out.cpu(), table[ids])
    assert int(err.item()) == 0
    bad = ids.clone()
    bad[1, 1] = 30522
    ops.embedding_gather(bad.to(DEV), table.to(DEV), out, err)
    assert int(err.item()) == 1 and torch.equal(out[1, 1].cpu(), torch.zeros(256))


# ------------------------------------------------------------------------------------------------ convolutions
GEOMS = [(3, 32, 65, 32, 64), (3, 32, 65, 64, 64), (2, 16, 32, 64, 128), (2, 16, 32, 128, 128),
         (2, 8, 16, 128, 256), (2, 8, 16, 256, 256), (32, 8, 16, 128, 256), (40, 16, 32, 64, 128)]


def test_weight_table_repack_is_a_permutation():
    """One launch re-lays-out a table of weight tensors (LDS-tiled): [Cout][Cin][taps] -> [tap][Cout][Cin] and
    [tap][Cin][Cout], bit for bit, for tile-aligned and ragged shapes, 1 to 9 taps, either output optional."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    shapes = [(256, 256, 9), (64, 32, 9), (64, 32, 1), (128, 256, 5), (128, 256, 4), (128, 256, 3), (40, 48, 9),
              (33, 31, 2), (96, 500, 3)]
    triples, refs = [], []
    for i, (co, ci, nt) in enumerate(shapes):
        w = torch.randn(co, ci, nt, generator=g).to(DEV)
        f = torch.full((w.numel(),), float("nan"), device=DEV) if i != 2 else None
        d = torch.full((w.numel(),), float("nan"), device=DEV) if i != 4 else None
        triples.append((w, f, d))
        refs.append((w.permute(2, 0, 1).contiguous().flatten(), w.permute(2, 1, 0).contiguous().flatten()))
    ops.PackPlan().run(triples)
    torch.cuda.synchronize()
    for (w, f, d), (rf, rd) in zip(triples, refs):
        assert f is None or torch.equal(f, rf), tuple(w.shape)
        assert d is None or torch.equal(d, rd), tuple(w.shape)
    with pytest.raises(ValueError):
        ops.PackPlan().run([(torch.zeros(4, 4, 10, device=DEV), torch.zeros(160, device=DEV), None)])


@pytest.mark.parametrize("B,H,W,Cin,Cout", GEOMS)
def test_conv3x3_forward_and_stats(B, H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    wf = torch.empty(w.numel(), device=DEV)
    ops.pack_weights(w.to(DEV), wf, None)
    y = torch.empty(B, H, W, Cout, device=DEV)
    rows, tiles = ops.conv_stats_rows(B, H, W, Cout)
    stats = torch.zeros(2, Cout, rows, device=DEV)
    ops.conv_fwd(nhwc(x).to(DEV), wf, y, stats, B, H, W, Cin, Cout, ops.TAPS3X3)
    check(nchw(y.cpu()), ref, 5e-6)
    s = stats.cpu().double().permute(2, 0, 1)      # [rows][2][Cout]
    refn = nhwc(ref)
    np.testing.assert_allclose(s[:, 0].sum(0).numpy(), refn.sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].sum(0).numpy(), (refn ** 2).sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-3)
    per_img = s[:, 0].view(B, tiles, Cout).sum(1)
    np.testing.assert_allclose(per_img.numpy(), refn.sum((1, 2)).numpy(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 32, 65, 32, 64), (2, 16, 31, 64, 128), (40, 8, 16, 128, 256)])
def test_paired_convolution_with_folded_batchnorm_relu(B, H, W, Cin, Cout):
    """A block's 3x3 convolution and 1x1 shortcut in one launch, the first stored as relu(scale * conv + shift) by the
    convolution's epilogue (inference): bit-identical to the plain launch followed by the BatchNorm + ReLU pass, the
    second result untouched; both against float64 convolutions.  Statistics cannot be asked for together with it."""
    ops = _ops()
    g = torch.Generator().manual_seed(5 * B + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w3 = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cout)) ** 0.5
    w1 = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    coeff = torch.randn(4, Cout, generator=g).to(DEV)
    xd = nhwc(x).to(DEV)
    wf3, wf1 = torch.empty(w3.numel(), device=DEV), torch.empty(w1.numel(), device=DEV)
    ops.pack_weights(w3.to(DEV), wf3, None)
    ops.pack_weights(w1.to(DEV), wf1, None)
    rows, _ = ops.conv_stats_rows(B, H, W, Cout)

    def run(fold, with_stats):
        y3 = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        y1 = torch.full((B, H, W, Cout), float("nan"), device=DEV)
        st = [torch.zeros(2, Cout, rows, device=DEV) for _ in range(2)] if with_stats else None
        ops.conv_fwd_multi(xd, [wf3, wf1], [y3, y1], [9, 1], B, H, W, Cin, Cout, ops.TAPS3X3, tap0_list=[0, 4],
                           stats_list=st, bnrelu_list=[coeff.flatten(), None] if fold else None)
        return y3, y1, st

    raw3, raw1, _ = run(False, False)
    two_pass = torch.empty_like(raw3)
    ops.bn_relu_apply(raw3, coeff.flatten(), two_pass, B * H * W, Cout)
    f3, f1, _ = run(True, False)
    assert torch.equal(f3, two_pass)
    assert torch.equal(f1, raw1)
    with pytest.raises(Exception):
        run(True, True)
    ref3 = F.conv2d(x.double(), w3.double(), None, 1, 1)
    ref = torch.relu(ref3 * coeff[2].cpu().double().view(1, -1, 1, 1) + coeff[3].cpu().double().view(1, -1, 1, 1))
    check(nchw(f3.cpu()), ref, 2e-5)
    check(nchw(f1.cpu()), F.conv2d(x.double(), w1.double()), 2e-6)
    with pytest.raises(ValueError):
        ops.conv_fwd_multi(xd, [wf3, wf1], [f3, f1], [9, 1], B, H, W, Cin, Cout, ops.TAPS3X3, tap0_list=[0, 4],
                           bnrelu_list=[coeff.flatten()[:Cout], None])


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 32, 65, 32, 64), (2, 16, 32, 64, 128), (2, 8, 16, 128, 256)])
def test_conv1x1_and_dgrad(B, H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator().manual_seed(7 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w3 = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    w1 = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    dy = torch.randn(B, Cout, H, W, generator=g)
    # 1x1 forward
    wf1, wd1 = torch.empty(w1.numel(), device=DEV), torch.empty(w1.numel(), device=DEV)
    ops.pack_weights(w1.to(DEV), wf1, wd1)
    y = torch.empty(B, H, W, Cout, device=DEV)
    ops.conv_fwd(nhwc(x).to(DEV), wf1, y, None, B, H, W, Cin, Cout, ops.TAPS1X1)
    check(nchw(y.cpu()), F.conv2d(x.double(), w1.double()), 2e-6)
    # data gradients: 3x3 then accumulate the 1x1 on top (what a residual block's backward does)
    wf3, wd3 = torch.empty(w3.numel(), device=DEV), torch.empty(w3.numel(), device=DEV)
    ops.pack_weights(w3.to(DEV), wf3, wd3)
    dx = torch.empty(B, H, W, Cin, device=DEV)
    dyd = nhwc(dy).to(DEV)
    ops.conv_fwd(dyd, wd3, dx, None, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w3.double(), None, 1, 1).backward(dy.double())
    ref3 = xr.grad.clone()
    check(nchw(dx.cpu()), ref3, 2e-6)
    ops.conv_fwd(dyd, wd1, dx, None, B, H, W, Cout, Cin, ops.TAPS1X1, negate=True, accumulate=True)
    xr.grad = None
    F.conv2d(xr, w1.double()).backward(dy.double())
    check(nchw(dx.cpu()), ref3 + xr.grad, 2e-6)
    # the same sum in one launch, with a different second gradient tensor (as in a residual block)
    dy2 = torch.randn(B, Cout, H, W, generator=g)
    dx2 = torch.empty(B, H, W, Cin, device=DEV)
    ops.conv_fwd_dual(dyd, wd3, nhwc(dy2).to(DEV), wd1, dx2, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    xr.grad = None
    F.conv2d(xr, w1.double()).backward(dy2.double())
    # up to K = 10 * 256 fp32 products per element, summed chunk by chunk: the rounding error of such a sum is of the
    # order sqrt(K) * 2^-24 = 3e-6 of the result's scale, whatever the order (measured 1.5e-6 .. 2.9e-6)
    check(nchw(dx2.cpu()), ref3 + xr.grad, 4e-6)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 32, 65, 32, 64), (3, 32, 65, 64, 64), (2, 16, 32, 64, 128),
                                            (2, 8, 16, 256, 256), (32, 8, 16, 128, 256)])
def test_conv_wgrad(B, H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator().manual_seed(11 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    dy = torch.randn(B, Cout, H, W, generator=g)
    for taps, k, pad in ((ops.TAPS3X3, 3, 1), (ops.TAPS1X1, 1, 0)):
        w = torch.zeros(Cout, Cin, k, k, dtype=torch.double, requires_grad=True)
        F.conv2d(x.double(), w, None, 1, pad).backward(dy.double())
        dW = torch.empty(Cout, Cin, k, k, device=DEV)
        wsz = ops.conv_wgrad_workspace(B, H, W, Cin, Cout, len(taps))
        wk = torch.empty(wsz, device=DEV)
        ops.conv_wgrad(nhwc(x).to(DEV), nhwc(dy).to(DEV), dW, wk, B, H, W, Cin, Cout, taps)
        check(dW.cpu(), w.grad, 5e-6)
        ops.conv_wgrad(nhwc(x).to(DEV), nhwc(dy).to(DEV), dW, wk, B, H, W, Cin, Cout, taps, accumulate=True)
        check(dW.cpu(), 2 * w.grad, 5e-6)


def test_text_conv_relu_maxtime():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, L, E, Fn = 5, 128, 256, 128
    emb = torch.randn(B, L, E, generator=g)
    feat = torch.zeros(B, 3 * Fn, device=DEV)
    arg = torch.zeros(B, 3 * Fn, dtype=torch.int32, device=DEV)
    refs, args = [], []
    for i, k in enumerate((3, 4, 5)):
        w = torch.randn(Fn, E, k, generator=g) * 0.05
        bias = torch.randn(Fn, generator=g) * 0.1
        wp = torch.empty(w.numel(), device=DEV)
        ops.pack_weights(w.to(DEV), wp, None)
        y = torch.empty(B, L, Fn, device=DEV)
        ops.conv_fwd(emb.to(DEV), wp, y, None, B, 1, L, E, Fn, [(0, j) for j in range(k)])
        ops.bias_relu_maxtime(y, bias.to(DEV), feat, arg, B, L, L - k + 1, Fn, 3 * Fn, i * Fn)
        c = F.relu(F.conv1d(emb.double().transpose(1, 2), w.double(), bias.double()))
        m, a = c.max(dim=2)
        refs.append(m)
        args.append(a)
    ref = torch.cat(refs, 1)
    check(feat.cpu(), ref, 2e-6)
    ra = torch.cat(args, 1)
    pos = ref > 1e-4          # where the max is a clear positive, the arg-max index must agree exactly
    assert torch.equal(arg.cpu().long()[pos], ra[pos])


@pytest.mark.parametrize("V", [3, 50, 1000])
def test_repeated_tokens_gradients_are_summed_in_position_order(V):
    """Embedding backward (autograd of multimodal_model.py:179-197): the table gradient is the same bits on every run,
    a token that occurs several times gets its rows added left to right in ascending position, the whole table agrees
    with float64 autograd, and the id table of the grouping kernel is left zero."""
    from mmsa_amd import text as text_engine
    from mmsa_amd.model import TextModel
    from mmsa_amd.workspace import Grads, Workspace
    torch.manual_seed(3)
    B, L, E = 16, 128, 256
    tm = TextModel(vocab_size=V).to(DEV)
    ids = torch.randint(0, V, (B, L), device=DEV)
    ids[:, 100:] = 0                      # padding tail
    ids[3, 5], ids[9, 0] = V + 7, -2      # outside the table: members of no group
    dfeat = torch.randn(B, 384, device=DEV)
    ws = Workspace(DEV, persistent=True)
    tables = []
    for _ in range(3):
        _, st = text_engine.text_forward(tm, ids, True, ws)
        grads = Grads()
        text_engine.text_backward(tm, st, dfeat, grads, ws)
        tables.append(grads.of(tm.embedding.weight).clone())
        assert int(ws.buf("text.id_table", 2 * V, torch.int32).abs().sum()) == 0
    st.err.zero_()
    assert torch.equal(tables[0], tables[1]) and torch.equal(tables[0], tables[2])
    drow = ws.buf("text.drow", (B * L, E)).cpu()
    flat = ids.flatten().cpu()
    got = tables[0].cpu()
    checked = 0
    for v in range(1, V):
        pos = (flat == v).nonzero().flatten().tolist()
        if len(pos) < 2:
            continue
        acc = drow[pos[0]].clone()
        for p in pos[1:]:
            acc = acc + drow[p]
        assert torch.equal(got[v], acc), v
        checked += 1
    assert checked > 0 and float(got[0].abs().max()) == 0.0
    # float64 autograd of the same graph (ids outside the table contribute nothing)
    w64 = tm.embedding.weight.detach().cpu().double().requires_grad_(True)
    safe = torch.where((ids >= 0) & (ids < V), ids, torch.zeros_like(ids)).cpu()
    x = F.embedding(safe, w64, padding_idx=0)
    bad = ((ids < 0) | (ids >= V)).cpu()
    x = torch.where(bad[..., None], torch.zeros_like(x), x).transpose(1, 2)
    feats = [F.relu(F.conv1d(x, c.weight.detach().cpu().double(), c.bias.detach().cpu().double())).max(dim=2)[0]
             for c in tm.convs]
    (torch.cat(feats, 1) * dfeat.cpu().double()).sum().backward()
    check(got, w64.grad, 2e-5)


# ------------------------------------------------------------------------------------------------ augmentation
def test_batched_augmentation_matches_oracle_draw_for_draw():
    """Same `random` seed -> the GPU batch path and the oracle's utterance-by-utterance restatement of
    preprocess.py:56-109,137-140,162-163 make the same decisions; with the GPU generator's own noise fed to the
    oracle the augmented waveforms are bit-identical, and the masked, normalised maps agree to 1e-4."""
    import random
    from mmsa_amd import ops
    from mmsa_amd.frontend import AudioTransform
    from oracle import augment_ref, frontend_ref
    B, n, seed = 24, 61000, 1234
    g = torch.Generator().manual_seed(5)
    waves = torch.randn(B, n, generator=g) * 0.1
    tf = AudioTransform(augment=True, device=DEV, noise_reduce=False)
    # the generator's N(0,1) draws per (utterance, sample): noise-only pass over silence
    only_noise = torch.tensor([[0.0, 0.0, 1.0, 0.0]] * B, dtype=torch.float64, device=DEV)
    z = torch.empty(B, n, device=DEV)
    ops.augment_wave(torch.zeros(B, n, device=DEV), only_noise, z, 1.0, seed)
    z = z.cpu()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.std()) - 1.0) < 3e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.05                       # Gaussian kurtosis
    z2 = torch.empty(B, n, device=DEV)
    ops.augment_wave(torch.zeros(B, n, device=DEV), only_noise, z2, 1.0, seed + 1)
    assert not torch.equal(z, z2.cpu())

    random.seed(77)
    want_w, want_m, used = [], [], set()
    for b in range(B):
        w, m, dropped = augment_ref.augment_utterance(waves[b].numpy(), lambda k, b=b: z[b, :k].numpy())
        want_w.append(w)
        want_m.append(m)
    random.seed(77)
    drawn = [tf.draw_augmentation(n) for _ in range(B)]
    rows, masks = [d[0] for d in drawn], [d[1] for d in drawn]
    assert masks == want_m
    for r in rows:
        used.add((r[0] != 0, r[2] != 0, r[3] != 0))
    assert len(used) >= 6                                                 # the sample exercises the combinations
    got_w = tf.augment_batch(waves, rows, masks, seed=seed, return_wave=True).cpu()
    bad = [(b, rows[b]) for b in range(B) if not torch.equal(got_w[b], torch.from_numpy(want_w[b]))]
    assert not bad, bad
    got = tf.augment_batch(waves, rows, masks, seed=seed).cpu()
    ref = frontend_ref.logmel_f64(np.stack(want_w), want_m)
    check(got, torch.from_numpy(ref), 1e-4)
    for b, m in enumerate(masks):                                         # masked cells share one value per map
        if m[1]:
            blk = got[b, 0, m[0]:m[0] + m[1], :]
            assert float(blk.max() - blk.min()) == 0.0


def test_zero_shift_drops_the_utterance_like_the_reference():
    """A drawn shift of 0 samples makes the reference return an all-zero map (its slice assignment raises)."""
    from mmsa_amd.frontend import AudioTransform
    tf = AudioTransform(augment=True, device=DEV)
    w = torch.randn(2, 30000) * 0.1
    rows = [[1.0, 0.0, 0.0, 0.0], [2.0, 3000.0, 0.0, 1.05]]
    out = tf.augment_batch(w, rows, [[0, 0, 0, 0]] * 2, seed=1).cpu()
    assert torch.equal(out[0], torch.zeros_like(out[0])) and float(out[1].abs().max()) > 0
    spec = tf(torch.randn(2, 20000) * 0.1, 16000)                          # per-utterance call, stereo 16 kHz
    assert spec.shape == (1, 64, 130) and bool(torch.isfinite(spec).all())


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(3, 8, 16, 128, 256), (2, 32, 65, 32, 64), (4, 16, 32, 64, 128)])
def test_paired_weight_gradients_equal_separate_launches(B, H, W, Cin, Cout):
    """The 3x3 and 1x1 weight gradients of one input computed in one launch are bit-identical to the two separate
    launch sequences (same strip plans, same fixed-order slab sums)."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * 1000 + W)
    X = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    dY9 = torch.randn(B, H, W, Cout, generator=g).to(DEV)
    dY1 = torch.randn(B, H, W, Cout, generator=g).to(DEV)
    ws9 = torch.empty(ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 9), device=DEV)
    ws1 = torch.empty(ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 1), device=DEV)
    a9, a1 = torch.empty(Cout, Cin, 3, 3, device=DEV), torch.empty(Cout, Cin, 1, 1, device=DEV)
    ops.conv_wgrad(X, dY9, a9, ws9, B, H, W, Cin, Cout, ops.TAPS3X3)
    ops.conv_wgrad(X, dY1, a1, ws1, B, H, W, Cin, Cout, ops.TAPS1X1)
    b9, b1 = torch.full_like(a9, 7.0), torch.full_like(a1, 7.0)
    ws9.fill_(3.0)
    ws1.fill_(3.0)
    ops.conv_wgrad_pair(X, dY9, b9, dY1, b1, ws9, ws1, B, H, W, Cin, Cout)
    assert torch.equal(a9, b9) and torch.equal(a1, b1)


def test_front_end_and_resampler_on_very_short_input():
    """Inputs shorter than one FFT frame / than the resampling filter: everything past them is zero padding."""
    from mmsa_amd.frontend import LogMelFrontEnd, Resampler
    from oracle import frontend_ref, resample_ref
    g = torch.Generator().manual_seed(9)
    tiny = torch.randn(2, 100, generator=g) * 0.1
    fe = LogMelFrontEnd(DEV)
    check(fe(tiny.to(DEV)).cpu(), torch.from_numpy(frontend_ref.logmel_f64(tiny.numpy())), 1e-4)
    pcm = torch.randn(1, 1, 10, generator=g) * 0.1
    got = Resampler(16000, 22050, DEV)(pcm.to(DEV)).cpu()
    ref = resample_ref.prepare(pcm[0].numpy(), 16000, dtype=np.float64)
    check(got, torch.from_numpy(ref[None]), 2e-6)
    assert torch.equal(got[:, 14:], torch.zeros(1, 66150 - 14))          # ceil(441 * 10 / 320) = 14 samples


def test_audio_transform_error_contract():
    """preprocess.py:172-174: a failing utterance yields an all-zero map; strict=True re-raises."""
    from mmsa_amd.frontend import AudioTransform
    bad = torch.zeros(2, 3, 100)                      # not [C, T]
    out = AudioTransform(device=DEV)(bad)
    assert out.shape == (1, 64, 130) and not out.any()
    with pytest.raises(Exception):
        AudioTransform(device=DEV, strict=True)(bad)


@pytest.mark.parametrize("n", [1, 255, 1024, 600100])
def test_noise_reduction_edge_lengths(n):
    """One sample, less than a hop, exactly one frame, and a signal one hundred samples longer than a chunk (its second
    chunk holds 100 samples and sees 30 000 samples of the first as context)."""
    from mmsa_amd.frontend import NoiseReducer
    from oracle import denoise_ref
    g = np.random.default_rng(n)
    y = (0.1 * g.standard_normal(n) + 0.2 * np.sin(np.arange(n) * 0.05) * (np.arange(n) > n // 2)).astype(np.float32)
    got = NoiseReducer(22050, DEV)(torch.from_numpy(y)[None].to(DEV)).cpu()[0]
    ref = denoise_ref.reduce_noise(y.astype(np.float64), 22050)
    assert got.shape == (n,) and bool(torch.isfinite(got).all())
    # relative to the input's peak: a one-sample signal is gated to (almost) nothing
    err = float((got.double() - torch.from_numpy(ref)).abs().max()) / float(np.abs(y).max())
    assert err < 5e-6, err


def test_split_k_dual_convolution_matches_the_unsplit_launch():
    """3x3 + 1x1 data gradient with the K range split over two workgroups per tile (atomic adds of two half-sums onto a
    zeroed map): equal to the one-workgroup launch up to the changed association of the sum, and bit-reproducible
    (two addends commute)."""
    ops = _ops()
    B, H, W, Cin, Cout = 32, 8, 16, 256, 128
    assert ops.conv_dual_prefers_split(B, H, W, Cout) and not ops.conv_dual_prefers_split(B, 32, 65, 64)
    g = torch.Generator().manual_seed(4)
    X = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    X2 = torch.randn(B, H, W, Cin, generator=g).to(DEV)
    Wp = (torch.randn(9, Cout, Cin, generator=g) * 0.05).to(DEV)
    Wp2 = (torch.randn(Cout, Cin, generator=g) * 0.05).to(DEV)
    ref = torch.empty(B, H, W, Cout, device=DEV)
    ops.conv_fwd_dual(X, Wp, X2, Wp2, ref, B, H, W, Cin, Cout, ops.TAPS3X3, negate=True)
    outs = []
    for _ in range(2):
        y = torch.zeros(B, H, W, Cout, device=DEV)
        ops.conv_fwd_dual_split(X, Wp, X2, Wp2, y, B, H, W, Cin, Cout, ops.TAPS3X3, negate=True)
        outs.append(y)
    check(outs[0], ref, 1e-5)          # fp32 sums over K = 2560 in two associations
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,H,W,C", [(3, 32, 65, 64), (2, 16, 32, 128), (5, 8, 16, 256), (2, 7, 19, 64)])
def test_data_gradient_with_batchnorm_backward_sums_in_its_epilogue(B, H, W, C):
    """The data gradient of a block's second convolution with the two reductions of bn1's backward formed in its
    epilogue, then mmsa_bn_relu_bwd_finish, against the same convolution followed by mmsa_bn_relu_bwd (which makes a
    pass of its own over the gradient and c1 for the sums), and against float64 autograd of relu(bn(c1)) -> conv."""
    ops = _ops()
    g = torch.Generator().manual_seed(3 + C)
    c1 = torch.randn(B, C, H, W, generator=g)
    w2 = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    d_c2 = torch.randn(B, C, H, W, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    # float64 reference: c1 -> BatchNorm (batch statistics) -> ReLU -> conv2, gradient d_c2 at the output
    x = c1.double().requires_grad_(True)
    ga, be = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    h = F.relu(F.batch_norm(x, None, None, ga, be, True, 0.1, 1e-5))
    F.conv2d(h, w2.double(), None, 1, 1).backward(d_c2.double())
    # the BatchNorm coefficient table [4][C]: mean, 1/std, scale, shift
    mean = c1.double().mean((0, 2, 3))
    var = c1.double().var((0, 2, 3), unbiased=False)
    istd = 1.0 / torch.sqrt(var + 1e-5)
    coeff = torch.stack([mean, istd, gamma.double() * istd, beta.double() - mean * gamma.double() * istd]).float().to(DEV)
    wd = torch.empty(w2.numel(), device=DEV)
    ops.pack_weights(w2.to(DEV), None, wd)
    c1d, dyd = nhwc(c1).to(DEV), nhwc(d_c2).to(DEV)
    npos = B * H * W
    outs = []
    for fused in (False, True):
        dh = torch.empty(B, H, W, C, device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        if fused:
            rows, _ = ops.conv_stats_rows(B, H, W, C)
            part = torch.zeros(2 * C * rows, device=DEV)
            ops.conv_fwd_bnrelu_bwd_stats(dyd, wd, dh, part, c1d, coeff.flatten(), B, H, W, C, C, ops.TAPS3X3, negate=True)
            ops.bn_relu_bwd_finish(dh, c1d, coeff.flatten(), part, rows, dg, db, torch.empty(3 * C, device=DEV), npos, C,
                                   True)
        else:
            ops.conv_fwd(dyd, wd, dh, None, B, H, W, C, C, ops.TAPS3X3, negate=True)
            ops.bn_relu_bwd(dh, c1d, coeff.flatten(), dg, db, torch.empty(ops.bn_relu_bwd_scratch(npos, C), device=DEV),
                            npos, C, True)
        outs.append((dh, dg, db))
        check(nchw(dh.cpu()), x.grad, 2e-5)
        check(dg.cpu(), ga.grad, 2e-5)
        check(db.cpu(), be.grad, 2e-5)
    for a, b in zip(*outs):
        check(a, b, 2e-6)


@pytest.mark.parametrize("B,H,W,C", [(4, 8, 16, 256), (3, 6, 10, 64), (2, 9, 7, 128)])
def test_global_average_pool_inside_the_last_blocks_pooling_kernels(B, H, W, C):
    """The last residual block's pooled output only feeds the global average pool: forward, the mean comes straight
    out of the gate + ReLU + max-pool kernel; backward, the block's element-wise backward takes the gradient of the
    mean.  Against the two-launch sequences (pooled map + gap_fwd, gap_bwd + gradient map)."""
    ops = _ops()
    g = torch.Generator().manual_seed(17 + C)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(DEV)
    c2, sc = rnd(B, H, W, C), rnd(B, H, W, C)
    coeff2, coeffs = rnd(4, C) * 0.5, rnd(4, C) * 0.5
    coeff2[1], coeffs[1] = coeff2[1].abs() + 0.5, coeffs[1].abs() + 0.5
    gate = torch.sigmoid(rnd(B, C))
    H2, W2 = H // 2, W // 2
    pooled = torch.empty(B, H2, W2, C, device=DEV)
    ops.res_gate_relu_pool(c2, sc, coeff2.flatten(), coeffs.flatten(), gate, pooled, B, H, W, C)
    feat_ref = torch.empty(B, C, device=DEV)
    ops.gap_fwd(pooled, feat_ref, B, H2 * W2, C)
    feat = torch.empty(B, C, device=DEV)
    ops.res_gate_relu_pool_mean(c2, sc, coeff2.flatten(), coeffs.flatten(), gate, feat, B, H, W, C)
    check(feat, pooled.double().mean((1, 2)), 2e-6)
    check(feat, feat_ref, 2e-6)
    # backward
    Ch = C // 16
    hid, gapv, bsum = rnd(B, Ch).abs(), rnd(B, C), rnd(B, C) * (H * W)
    W1, W2m = rnd(Ch, C) * 0.1, rnd(C, Ch) * 0.1
    dfeat = rnd(B, C)
    dmap = torch.empty(B, H2 * W2, C, device=DEV)
    ops.gap_bwd(dfeat, dmap, B, H2 * W2, C)
    outs = []
    for mean in (False, True):
        d_c2, d_sc = torch.empty_like(c2), torch.empty_like(c2)
        gr = [torch.zeros(C, device=DEV) for _ in range(4)]
        dW1, dW2 = torch.zeros_like(W1), torch.zeros_like(W2m)
        scratch = torch.zeros(ops.resblock_bwd_scratch(B, H, W, C), device=DEV)
        ops.resblock_bwd_ew(dfeat if mean else dmap, c2, sc, coeff2.flatten(), coeffs.flatten(), gate, hid, gapv, bsum,
                            W1, W2m, d_c2, d_sc, gr[0], gr[1], gr[2], gr[3], dW1, dW2, scratch, B, H, W, C, True,
                            dout_is_mean=mean)
        outs.append([d_c2, d_sc, *gr, dW1, dW2])
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_both_implicit_gemm_kernels_on_every_launch_shape(monkeypatch):
    """Which kernel takes a 3x3 launch (MMSA_IGEMM3 = 0: the general implicit GEMM always; 1: the one that stages an
    input strip once per row of three taps for 32-channel results only; 2, the default: for every 3x3 launch), and
    both kernels against the same float64 references: statistics epilogue, multi-problem launch (3x3 + centre tap,
    folded BatchNorm + ReLU), second operand pair, accumulate flag, split K range."""
    ops = _ops()
    from mmsa_amd import _lib
    dh, dw = [t[0] for t in ops.TAPS3X3], [t[1] for t in ops.TAPS3X3]

    def shares(B, H, W, Cout):
        return _lib.query("mmsa_conv_shares_tap_rows", B, H, W, Cout, 9, ops._iarr(dh), ops._iarr(dw))

    monkeypatch.setenv("MMSA_IGEMM3", "1")
    assert shares(32, 32, 65, 32) == 1 and shares(32, 32, 65, 64) == 0
    monkeypatch.delenv("MMSA_IGEMM3", raising=False)
    assert shares(32, 32, 65, 32) == 1 and shares(32, 32, 65, 64) == 1 and shares(32, 8, 16, 256) == 1
    test_conv3x3_forward_and_stats(2, 5, 3, 32, 64)        # three columns: every tap row wraps within the tile
    test_conv3x3_forward_and_stats(1, 1, 200, 32, 64)      # one image row: the rows above and below are all padding
    test_conv3x3_forward_and_stats(2, 1, 1, 32, 64)        # a single pixel
    monkeypatch.setenv("MMSA_IGEMM3", "0")                 # the general kernel on the same cases
    assert shares(32, 32, 65, 32) == 0
    for geom in GEOMS:
        test_conv3x3_forward_and_stats(*geom)
    for geom in [(3, 32, 65, 32, 64), (2, 16, 31, 64, 128), (40, 8, 16, 128, 256)]:
        test_paired_convolution_with_folded_batchnorm_relu(*geom)
    for geom in [(3, 32, 65, 32, 64), (2, 16, 32, 64, 128), (2, 8, 16, 128, 256)]:
        test_conv1x1_and_dgrad(*geom)
    test_split_k_dual_convolution_matches_the_unsplit_launch()


@pytest.mark.parametrize("B,H,W,Cin,Cout,sc", [(3, 8, 16, 256, 256, False), (2, 8, 16, 128, 256, True),
                                                 (3, 16, 32, 128, 128, False), (2, 16, 32, 64, 128, True),
                                                 (3, 32, 65, 64, 64, False), (2, 32, 65, 32, 64, True),
                                                 (1, 5, 7, 32, 64, True), (9, 3, 33, 64, 64, False)])
def test_weight_gradient_slabs_match_float64(B, H, W, Cin, Cout, sc):
    """The DMA-fed weight-gradient kernel + the multi-layer slab reduction against float64 autograd of F.conv2d, for
    the six layer shapes of the model (shortcut riding along where the model has one) and two ragged geometries (odd
    width, height that no strip height divides, fewer strips than workgroups)."""
    from mmsa_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + W)
    x = torch.randn(B, H, W, Cin, generator=g)
    dy = torch.randn(B, H, W, Cout, generator=g)
    dy1 = torch.randn(B, H, W, Cout, generator=g) if sc else None
    xd = x.permute(0, 3, 1, 2).double()
    w9 = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, w9, padding=1).backward(dy.permute(0, 3, 1, 2).double())
    ns = ops.conv_wgrad_slab_count(B, H, W, Cin, Cout, sc)
    assert 1 <= ns <= 256
    slabs = torch.full((ns, 9, Cin, Cout), float("nan"), device=DEV)
    slabs1 = torch.full((ns, Cin, Cout), float("nan"), device=DEV) if sc else None
    got_ns = ops.conv_wgrad_slabs(x.to(DEV), dy.to(DEV), None if dy1 is None else dy1.to(DEV), slabs, slabs1,
                                  B, H, W, Cin, Cout)
    assert got_ns == ns
    dW9 = torch.full((Cout, Cin, 3, 3), float("nan"), device=DEV)
    entries = [(slabs, dW9, ns, 9, Cin, Cout)]
    if sc:
        dW1 = torch.full((Cout, Cin, 1, 1), float("nan"), device=DEV)
        entries.append((slabs1, dW1, ns, 1, Cin, Cout))
    plan = ops.ReducePlan()
    plan.run(entries)
    plan.run(entries)                      # the cached table
    scale = float(w9.grad.abs().max())
    assert float((dW9.cpu().double() - w9.grad).abs().max()) < 2e-5 * scale
    if sc:
        w1 = torch.zeros(Cout, Cin, 1, 1, dtype=torch.float64, requires_grad=True)
        F.conv2d(xd, w1).backward(dy1.permute(0, 3, 1, 2).double())
        assert float((dW1.cpu().double() - w1.grad).abs().max()) < 2e-5 * float(w1.grad.abs().max())


@pytest.mark.parametrize("B,H,W,C,mean", [(3, 32, 65, 64, False), (5, 16, 32, 128, False), (2, 7, 19, 64, False),
                                          (4, 8, 16, 256, True), (33, 8, 16, 256, False), (3, 6, 10, 64, True)])
def test_squeeze_excite_gate_inside_the_pooling_launch(B, H, W, C, mean):
    """The gate worked out by every workgroup of the pooling launch for its utterance against the separate gate launch
    followed by the pooling launch: pooled map (or its mean), gap, hidden units and gate bit for bit."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(C + W)
    Ch = C // 16
    c2 = torch.randn(B, H, W, C, generator=g, device=DEV)
    sc = torch.randn(B, H, W, C, generator=g, device=DEV)
    coeff2 = torch.randn(4, C, generator=g, device=DEV)
    coeffs = torch.randn(4, C, generator=g, device=DEV)
    bsum = torch.randn(B, C, generator=g, device=DEV) * H * W
    W1 = torch.randn(Ch, C, generator=g, device=DEV) * 0.2
    W2 = torch.randn(C, Ch, generator=g, device=DEV) * 0.5
    ref = [torch.empty(B, C, device=DEV), torch.empty(B, Ch, device=DEV), torch.empty(B, C, device=DEV)]
    ops.se_gate(bsum, coeff2, W1, W2, H * W, ref[0], ref[1], ref[2], B, C)
    got = [torch.full_like(t, float("nan")) for t in ref]
    if mean:
        want = torch.empty(B, C, device=DEV)
        ops.res_gate_relu_pool_mean(c2, sc, coeff2, coeffs, ref[2], want, B, H, W, C)
        out = torch.full_like(want, float("nan"))
        ops.res_se_gate_relu_pool(c2, sc, coeff2, coeffs, bsum, W1, W2, got[0], got[1], got[2], None, out, B, H, W, C)
    else:
        want = torch.empty(B, H // 2, W // 2, C, device=DEV)
        ops.res_gate_relu_pool(c2, sc, coeff2, coeffs, ref[2], want, B, H, W, C)
        out = torch.full_like(want, float("nan"))
        ops.res_se_gate_relu_pool(c2, sc, coeff2, coeffs, bsum, W1, W2, got[0], got[1], got[2], out, None, B, H, W, C)
    assert torch.equal(out, want)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)


def test_operands_on_another_device_than_the_current_one_are_refused(monkeypatch):
    """Kernels launch on the current device.  A process that drives several GPUs and hands over tensors of a device
    that is not current would run a kernel on the wrong GPU: the binding refuses by name instead.  (One GPU here: the
    current device is faked.)  The model's forward and TrainStep.step make their device current themselves."""
    from mmsa_amd import _lib, ops
    x = torch.zeros(4, 8, device=DEV)
    ops._chk(x)
    assert ops._stream() == torch.cuda.current_stream().cuda_stream
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    ops._chk(x)
    with pytest.raises(_lib.MmsaError, match="current device"):
        ops._stream()

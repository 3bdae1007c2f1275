"""Properties that hold at any size, checked at the sizes the headline metric is quoted on (batch 32 per GPU, the three
residual blocks' maps, the 30 522-row embedding table, the 38.5 MB flat parameter buffer), where the CPU oracle would
take minutes.  Nothing here needs an oracle: the kernels are checked against each other (a convolution, its data
gradient and its weight gradient are one bilinear form: <conv(X, W), Y> = <X, dgrad(Y, W)> = <W, wgrad(X, Y)>),
against linearity, and against plain torch arithmetic on the device for the element-wise optimiser.  Small-size
parity against the oracle is in test_gpu_ops.py / test_gpu_model.py / test_gpu_train.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
B = 32
# (H, W, Cin, Cout) of the six 3x3 convolutions of the audio branch at the bench's input (64 x 130 log-mel map)
LAYERS = [(32, 65, 32, 64), (32, 65, 64, 64), (16, 32, 64, 128), (16, 32, 128, 128), (8, 16, 128, 256),
          (8, 16, 256, 256)]


def _ops():
    from mmsa_amd import ops
    return ops


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("H,W,Cin,Cout", LAYERS)
def test_convolution_data_gradient_and_weight_gradient_are_one_bilinear_form(H, W, Cin, Cout):
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(H * 1000 + Cin)
    x = torch.randn(B, H, W, Cin, generator=g, device=DEV)
    y = torch.randn(B, H, W, Cout, generator=g, device=DEV)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device=DEV) * (2.0 / (9 * Cin)) ** 0.5
    wf, wd = torch.empty(w.numel(), device=DEV), torch.empty(w.numel(), device=DEV)
    ops.pack_weights(w, wf, wd)
    out = torch.empty(B, H, W, Cout, device=DEV)
    rows, tiles = ops.conv_stats_rows(B, H, W, Cout)
    stats = torch.zeros(2, Cout, rows, device=DEV)
    ops.conv_fwd(x, wf, out, stats, B, H, W, Cin, Cout, ops.TAPS3X3)
    dx = torch.empty(B, H, W, Cin, device=DEV)
    ops.conv_fwd(y, wd, dx, None, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    ns = ops.conv_wgrad_slab_count(B, H, W, Cin, Cout, False)
    slabs = torch.empty(ns, 9, Cin, Cout, device=DEV)
    ops.conv_wgrad_slabs(x, y, None, slabs, None, B, H, W, Cin, Cout)
    dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    ops.ReducePlan().run([(slabs, dw, ns, 9, Cin, Cout)])
    torch.cuda.synchronize()
    f, d, wg = _dot(out, y), _dot(x, dx), _dot(w, dw)
    # each side is a sum of B*H*W*Cout*9*Cin products of fp32-rounded partial results; the three agree to the rounding
    # of those partial results (1e-6 of the typical magnitude sqrt(N) of such a sum), far below what a wrong tap,
    # halo, tile edge or split would cause (a single mis-placed tile changes the sum by 1e-3 of it)
    scale = (out.double().pow(2).sum() * y.double().pow(2).sum()).sqrt().item() / out.numel() ** 0.5
    assert abs(f - d) < 2e-5 * scale and abs(f - wg) < 2e-5 * scale, (f, d, wg, scale)
    # the statistics epilogue against the map it rode on
    s = stats.double().permute(2, 0, 1)
    assert torch.allclose(s[:, 0].sum(0), out.double().sum((0, 1, 2)), rtol=1e-5, atol=1e-2)
    assert torch.allclose(s[:, 1].sum(0), out.double().pow(2).sum((0, 1, 2)), rtol=1e-5, atol=1e-2)
    # linearity in the input: conv(2.5 x + x2) = 2.5 conv(x) + conv(x2)
    x2 = torch.randn(B, H, W, Cin, generator=g, device=DEV)
    o2, o3 = torch.empty_like(out), torch.empty_like(out)
    ops.conv_fwd(x2, wf, o2, None, B, H, W, Cin, Cout, ops.TAPS3X3)
    ops.conv_fwd(2.5 * x + x2, wf, o3, None, B, H, W, Cin, Cout, ops.TAPS3X3)
    err = float((o3.double() - (2.5 * out.double() + o2.double())).abs().max())
    assert err < 2e-5 * float(o3.abs().max()), err


@pytest.mark.parametrize("H,W,Cin,Cout", [(32, 65, 32, 64), (16, 32, 64, 128), (8, 16, 128, 256)])
def test_fused_data_gradient_and_shortcut_weight_gradient_at_full_size(H, W, Cin, Cout):
    """A block's first convolution and its 1x1 shortcut: the fused (and, on the small maps, K-split) data gradient
    against the two separate launches, and the shortcut's weight gradient riding on the 3x3 launch against its own
    bilinear form."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(W * 77 + Cout)
    x = torch.randn(B, H, W, Cin, generator=g, device=DEV)
    d3 = torch.randn(B, H, W, Cout, generator=g, device=DEV)
    d1 = torch.randn(B, H, W, Cout, generator=g, device=DEV)
    w3 = torch.randn(Cout, Cin, 3, 3, generator=g, device=DEV) * 0.05
    w1 = torch.randn(Cout, Cin, 1, 1, generator=g, device=DEV) * 0.1
    f3, b3 = torch.empty(w3.numel(), device=DEV), torch.empty(w3.numel(), device=DEV)
    f1, b1 = torch.empty(w1.numel(), device=DEV), torch.empty(w1.numel(), device=DEV)
    ops.pack_weights(w3, f3, b3)
    ops.pack_weights(w1, f1, b1)
    sep = torch.empty(B, H, W, Cin, device=DEV)
    ops.conv_fwd(d3, b3, sep, None, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    ops.conv_fwd(d1, b1, sep, None, B, H, W, Cout, Cin, ops.TAPS1X1, negate=True, accumulate=True)
    dual = torch.empty(B, H, W, Cin, device=DEV)
    ops.conv_fwd_dual(d3, b3, d1, b1, dual, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    scale = float(sep.abs().max())
    assert float((dual - sep).abs().max()) < 4e-6 * scale
    if ops.conv_dual_prefers_split(B, H, W, Cin):
        split = torch.zeros(B, H, W, Cin, device=DEV)
        ops.conv_fwd_dual_split(d3, b3, d1, b1, split, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
        assert float((split - sep).abs().max()) < 4e-6 * scale
    ns = ops.conv_wgrad_slab_count(B, H, W, Cin, Cout, True)
    slabs = torch.empty(ns, 9, Cin, Cout, device=DEV)
    slabs1 = torch.empty(ns, Cin, Cout, device=DEV)
    ops.conv_wgrad_slabs(x, d3, d1, slabs, slabs1, B, H, W, Cin, Cout)
    dw3 = torch.empty(Cout, Cin, 3, 3, device=DEV)
    dw1 = torch.empty(Cout, Cin, 1, 1, device=DEV)
    ops.ReducePlan().run([(slabs, dw3, ns, 9, Cin, Cout), (slabs1, dw1, ns, 1, Cin, Cout)])
    y3, y1 = torch.empty(B, H, W, Cout, device=DEV), torch.empty(B, H, W, Cout, device=DEV)
    ops.conv_fwd(x, f3, y3, None, B, H, W, Cin, Cout, ops.TAPS3X3)
    ops.conv_fwd(x, f1, y1, None, B, H, W, Cin, Cout, ops.TAPS1X1)
    for out, dy, w, dw in ((y3, d3, w3, dw3), (y1, d1, w1, dw1)):
        sc = (out.double().pow(2).sum() * dy.double().pow(2).sum()).sqrt().item() / out.numel() ** 0.5
        assert abs(_dot(out, dy) - _dot(w, dw)) < 2e-5 * sc


def test_embedding_gather_at_the_full_vocabulary_is_bit_exact():
    from mmsa_amd import ops
    g = torch.Generator(device=DEV).manual_seed(3)
    table = torch.randn(30522, 256, generator=g, device=DEV)
    ids = torch.randint(0, 30522, (B, 128), generator=g, device=DEV)
    ids[:, 0], ids[:, -1], ids[3, 5] = 101, 102, 30521
    out = torch.full((B, 128, 256), float("nan"), device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embedding_gather(ids, table, out, err)
    assert torch.equal(out, table[ids]) and int(err) == 0


def test_clip_and_adamw_over_the_models_flat_buffer_follow_the_formulas():
    """One fused clip + AdamW step over the 'both' model's whole parameter buffer (9.6 M floats, 71 tensors) against
    the same formulas written with torch operations on the device: global norm of g + 0.001 p / ||p|| per tensor,
    coefficient min(1, 1 / (norm + 1e-6)), decoupled weight decay, bias corrections of step 1."""
    from mmsa_amd import ops
    from mmsa_amd.model import L2_LAMBDA, MultimodalSentimentModel
    torch.manual_seed(0)
    m = MultimodalSentimentModel("both").to(DEV)
    a = m._ensure_arena()
    gen = torch.Generator(device=DEV).manual_seed(9)
    g = torch.zeros_like(a.flat)
    for v in a.views_of(g):
        v.copy_(torch.randn(v.shape, generator=gen, device=DEV) * 1e-2)
    p0 = a.flat.clone()
    mom, var, state = torch.zeros_like(p0), torch.zeros_like(p0), torch.zeros(8, device=DEV)
    psum = torch.zeros(1, device=DEV)
    ops.param_norms(a.flat, a.table, a.tnorm, psum)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-4
    ops.grad_norm_finish(g, a.table, state, b1, b2, 1.0, True, a.flat, a.tnorm, L2_LAMBDA)
    ops.adamw_step(a.flat, g, mom, var, a.table, state, lr, b1, b2, eps, wd, 1.0, True, a.tnorm, L2_LAMBDA,
                   emit_param_sums=True, scalars_ready=True)
    torch.cuda.synchronize()
    full = torch.zeros_like(p0, dtype=torch.float64)
    for gv, pv, fv in zip(a.views_of(g), a.views_of(p0), a.views_of(full)):
        n = float(pv.double().norm())              # zero-initialised biases: torch's norm has gradient 0 there
        fv.copy_(gv.double() + (L2_LAMBDA * pv.double() / n if n > 0 else 0.0))
    norm = float(full.norm())
    assert abs(float(state[4]) - norm) < 1e-5 * norm
    gc = full * min(1.0, 1.0 / (norm + 1e-6))
    m1 = (1 - b1) * gc
    v1 = (1 - b2) * gc * gc
    upd = (m1 / (1 - b1)) / ((v1 / (1 - b2)).sqrt() + eps)
    want = p0.double() * (1 - lr * wd) - lr * upd
    used = torch.zeros_like(p0, dtype=torch.bool)
    for uv in a.views_of(used):
        uv.fill_(True)
    # an element whose clipped gradient is of the order of eps or of fp32 rounding takes a step anywhere in +-lr:
    # only elements with a gradient well above that are held to rounding; all are held to the step bound
    diff = (a.flat.double() - want).abs()
    solid = used & (gc.abs() > 1e-5)
    # (the new value is an fp32 number: half a unit in its last place, plus the step's own rounding)
    bar = 1.2e-7 * want.abs() + 2e-6 * lr
    assert bool((diff[solid] < bar[solid]).all()) and int(solid.sum()) > 9_000_000, float((diff - bar)[solid].max())
    assert float(diff[used].max()) <= 2.0 * lr and float(diff[~used].max()) == 0.0     # gaps stay untouched
    assert torch.allclose(mom.double()[used], m1[used], rtol=1e-5, atol=1e-12)
    # (1 - beta2 reaches the kernel as the fp32 rounding of the double difference, 0.001f, as in torch; formed in fp32
    # from the fp32 beta2 it would be 0.00099998713, 1.3e-5 off: this bar tells the two apart)
    assert torch.allclose(var.double()[used], v1[used], rtol=2e-6, atol=1e-16)


def test_front_end_output_is_standardised_per_utterance_at_batch_32():
    """Whatever the waveform, every utterance's log-mel map leaves the front end with mean 0 and unbiased standard
    deviation 1 up to the clamp at +-6 (preprocess.py:111-123), and finite."""
    from mmsa_amd import data
    from mmsa_amd.frontend import LogMelFrontEnd
    b = data.make_batch(0, B, with_wave=True)
    wave = torch.from_numpy(b["wave"]).to(DEV)
    spec = LogMelFrontEnd(DEV)(wave)
    assert spec.shape == (B, 1, 64, 130) and bool(torch.isfinite(spec).all())
    flat = spec.double().flatten(1)
    clamped = (flat.abs() >= 6.0).double().mean(1)
    ok = clamped == 0
    assert int(ok.sum()) >= B // 2                               # the statement is exact where nothing was clamped
    assert float(flat[ok].mean(1).abs().max()) < 1e-5 and float((flat[ok].std(1, unbiased=True) - 1).abs().max()) < 1e-5
    assert float(flat.abs().max()) <= 6.0


def test_norms_over_more_chunks_than_the_norm_kernel_stages():
    """Per-tensor norms and the global gradient norm over a flat buffer of 75 M floats in three tensors (4 600
    16 K-element chunks: more than the one-workgroup norm kernel keeps in LDS, so it reads the chunk sums from memory),
    and over a small one (staged), against torch's norms."""
    from mmsa_amd import ops
    from mmsa_amd.arena import ChunkTable
    gen = torch.Generator(device=DEV).manual_seed(3)
    for sizes in ([70_000_000, 5_000_003, 17], [40_000, 3, 16_385]):
        offsets, pos = [], 0
        for n in sizes:
            offsets.append(pos)
            pos += (n + 255) // 256 * 256
        table = ChunkTable(offsets, sizes, DEV)
        assert (table.nchunks > 4096) == (sizes[0] > 10 ** 6)
        g = torch.randn(pos, generator=gen, device=DEV)
        tnorm = torch.zeros(len(sizes), device=DEV)
        psum = torch.zeros(1, device=DEV)
        ops.param_norms(g, table, tnorm, psum)
        state = torch.zeros(8, device=DEV)
        ops.grad_norm_finish(g, table, state, 0.9, 0.999, 1.0, True)
        torch.cuda.synchronize()
        want = torch.stack([g[o:o + n].double().norm() for o, n in zip(offsets, sizes)])
        assert float((tnorm.double() - want).abs().max() / want.max()) < 1e-6
        assert abs(float(psum) - float(want.sum())) < 1e-5 * float(want.sum())
        total = float(want.square().sum().sqrt())
        assert abs(float(state[4]) - total) < 1e-6 * total
        assert float(state[0]) == 1.0 and abs(float(state[3]) - min(1.0, 1.0 / (total + 1e-6))) < 1e-6

"""The entry point that allocates nothing (mmsa_ln_scale_bwd_static: scratch in a device array inside the library)
against the entry point it is a variant of (mmsa_ln_scale_bwd, scratch from the caller), through the C ABI."""
import pytest
import torch

from conftest import SMALL_VOCAB  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(B, N, seed, keep):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)   # noqa: E731
    x, dy = r(B, N), r(B, N + 8)[:, :N]                   # dy rows with a leading dimension larger than N
    gamma, beta, scale = 1.0 + 0.1 * r(N), 0.1 * r(N), 1.0 + 0.1 * r(1)
    kp = torch.full((1,), keep, device=DEV) if keep is not None else None
    return x, dy, gamma, beta, scale, kp


def _run(entry, x, dy, gamma, beta, scale, kp, scratch=None):
    from mmsa_amd import _lib, ops
    B, N = x.shape
    y = torch.empty(B, N, device=DEV)
    stats = torch.empty(2 * B, device=DEV)
    ops.ln_scale_fwd(x, gamma, beta, scale, kp, y, stats)
    dx = torch.full((B, N), float("nan"), device=DEV)
    dgamma, dbeta, dscale = (torch.full((n,), float("nan"), device=DEV) for n in (N, N, 1))
    args = [dy.data_ptr(), dy.stride(0), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), scale.data_ptr(),
            None if kp is None else kp.data_ptr(), stats.data_ptr(), dx.data_ptr(), dgamma.data_ptr(),
            dbeta.data_ptr(), dscale.data_ptr()]
    if scratch is not None:
        args.append(scratch.data_ptr())
    _lib.call(entry, *args, B, N, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dx, dgamma, dbeta, dscale


def test_static_scratch_entry_point_equals_the_workspace_one_twice_in_a_row():
    from mmsa_amd import _lib
    # two calls in a row, the second with a smaller input: whatever the first call left in the library's array must
    # not leak into the second result (bit-exact against the variant with caller scratch, which starts from NaN)
    for B, N, seed, keep in ((64, 384, 1, 1.0), (5, 256, 2, None), (1, 40, 3, 0.0)):
        case = _case(B, N, seed, keep)
        scratch = torch.full((B * N + B,), float("nan"), device=DEV)
        want = _run("mmsa_ln_scale_bwd", *case, scratch=scratch)
        got = _run("mmsa_ln_scale_bwd_static", *case)
        for w, g_ in zip(want, got):
            assert bool(torch.isfinite(w).all())
            assert torch.equal(w, g_)
    # float64 check of the gradient itself (1e-5 relative): LayerNorm(x) * scale * keep
    x, dy, gamma, beta, scale, kp = _case(7, 384, 4, 1.0)
    dx, dgamma, dbeta, dscale = _run("mmsa_ln_scale_bwd_static", x, dy, gamma, beta, scale, kp)
    xd = x.double().requires_grad_(True)
    gd, bd, sd = (t.double().requires_grad_(True) for t in (gamma, beta, scale))
    yd = torch.nn.functional.layer_norm(xd, (384,), gd, bd, 1e-5) * sd
    yd.backward(dy.double())
    for got, ref in ((dx, xd.grad), (dgamma, gd.grad), (dbeta, bd.grad), (dscale, sd.grad)):
        assert float((got.double() - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))
    # beyond the array's size the call is refused and nothing is launched
    B, N = 4096, 384
    big = torch.zeros(B, N, device=DEV)
    v = torch.zeros(N, device=DEV)
    one = torch.ones(1, device=DEV)
    st = torch.zeros(2 * B, device=DEV)
    rc = _lib.load().mmsa_ln_scale_bwd_static(big.data_ptr(), N, big.data_ptr(), v.data_ptr(), v.data_ptr(),
                                              one.data_ptr(), None, st.data_ptr(), big.data_ptr(), v.data_ptr(),
                                              v.data_ptr(), one.data_ptr(), B, N, None)
    assert rc == -3
    with pytest.raises(_lib.MmsaError, match="limit"):
        _lib.call("mmsa_ln_scale_bwd_static", big.data_ptr(), N, big.data_ptr(), v.data_ptr(), v.data_ptr(),
                  one.data_ptr(), None, st.data_ptr(), big.data_ptr(), v.data_ptr(), v.data_ptr(), one.data_ptr(),
                  B, N, None)


def test_static_scratch_entry_point_under_graph_capture():
    """What the entry point is for: a captured launch sequence with no workspace of its own."""
    case = _case(16, 256, 9, 1.0)
    want = _run("mmsa_ln_scale_bwd", *case, scratch=torch.empty(16 * 256 + 16, device=DEV))
    from mmsa_amd import _lib, ops
    x, dy, gamma, beta, scale, kp = case
    B, N = x.shape
    stats = torch.empty(2 * B, device=DEV)
    ops.ln_scale_fwd(x, gamma, beta, scale, kp, torch.empty(B, N, device=DEV), stats)
    dx = torch.zeros(B, N, device=DEV)
    dgamma, dbeta, dscale = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV), torch.zeros(1, device=DEV)

    def launch():
        _lib.call("mmsa_ln_scale_bwd_static", dy.data_ptr(), dy.stride(0), x.data_ptr(), gamma.data_ptr(),
                  beta.data_ptr(), scale.data_ptr(), kp.data_ptr(), stats.data_ptr(), dx.data_ptr(), dgamma.data_ptr(),
                  dbeta.data_ptr(), dscale.data_ptr(), B, N, torch.cuda.current_stream().cuda_stream)
    # the per-device look-up of the array's address is a runtime query: done outside the capture, by the init call
    _lib.call("mmsa_static_scratch_init")
    _lib.call("mmsa_static_scratch_init")      # idempotent
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        launch()
    for t in (dx, dgamma, dbeta, dscale):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    for w, got in zip(want, (dx, dgamma, dbeta, dscale)):
        assert torch.equal(w, got)

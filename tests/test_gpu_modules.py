"""Stand-alone forwards of the drop-in sub-modules (`CNN implementation/multimodal_model.py:5-70,202-285`) against the
CPU oracle, and the torch.ops.mmsa registration they go through."""
import pytest
import torch

from conftest import SMALL_VOCAB  # noqa: F401  (puts the package on sys.path)
from oracle import model_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _sd(module, prefix):
    return {f"{prefix}.{k}": v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _close(got, want, tol=1e-4):
    want = want.float()
    assert got.shape == want.shape
    err = float((got.cpu() - want).abs().max())
    assert err < tol * max(1.0, float(want.abs().max())), err


def _grad_close(got, want, tol=2e-4):
    assert got is not None and want is not None and got.shape == want.shape
    err = float((got.detach().cpu().double() - want.double()).abs().max())
    assert err <= tol * float(want.abs().max()) + 1e-6, (err, float(want.abs().max()))   # (1e-6: gradients that are rounding noise around an analytic zero)


def test_operators_are_registered_and_have_no_cpu_path():
    from mmsa_amd import torch_ops
    assert len(torch_ops.REGISTERED) >= 25
    for name in ("conv_fwd", "se_gate", "gate_affine_add", "ln_scale_fwd", "dense_fwd", "fusion_mix_fwd", "logmel",
                 "embedding_gather", "conv_wgrad_slabs"):
        assert hasattr(torch.ops.mmsa, name)
    with pytest.raises(NotImplementedError):
        torch.ops.mmsa.gap_fwd(torch.zeros(2, 4, 8), torch.zeros(2, 8), 2, 4, 8)
    # an operator and the launcher it wraps write the same bytes
    from mmsa_amd import ops
    x = torch.randn(3, 20, 64, device=DEV)
    a, b = torch.empty(3, 64, device=DEV), torch.empty(3, 64, device=DEV)
    torch.ops.mmsa.gap_fwd(x, a, 3, 20, 64)
    ops.gap_fwd(x, b, 3, 20, 64)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C,red", [(64, 16), (256, 16), (128, 8)])
def test_se_block_alone(C, red):
    from mmsa_amd.model import SEBlock
    torch.manual_seed(C)
    m = SEBlock(C, red).to(DEV)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.3)
    x = torch.randn(3, C, 8, 10)
    with torch.no_grad():
        got = m(x.to(DEV))
    _close(got, model_ref.se_block(_sd(m, "se"), "se", x))
    # autograd through the block on its own: input and weight gradients against the oracle's
    sd = model_ref.clone_state(_sd(m, "se"), requires_grad=True)
    xr = x.clone().requires_grad_(True)
    R = torch.randn(3, C, 8, 10)
    (model_ref.se_block(sd, "se", xr) * R).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    (m(xg) * R.to(DEV)).sum().backward()
    _grad_close(xg.grad, xr.grad)
    for k, p in m.named_parameters():
        _grad_close(p.grad, sd["se." + k].grad)


@pytest.mark.parametrize("cin,cout,training", [(32, 64, True), (64, 128, False), (64, 64, True), (128, 128, False)])
def test_residual_block_alone(cin, cout, training):
    """Projection shortcut (channels change) and identity shortcut (they do not), batch statistics and running
    statistics; in training mode the running statistics must advance as the reference's BatchNorm layers do."""
    from mmsa_amd.model import ResidualBlock
    torch.manual_seed(cin + cout)
    m = ResidualBlock(cin, cout).to(DEV).train(training)
    assert (len(m.shortcut) == 0) == (cin == cout)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "se.fc" in k:
                p.normal_(0, 0.3)
        for k, b in m.named_buffers():
            if k.endswith("running_mean"):
                b.normal_(0, 0.2)
            elif k.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    sd = _sd(m, "rb")
    x = torch.randn(3, cin, 8, 12)
    want = model_ref.residual_block(sd, "rb", x, training)
    with torch.no_grad():
        got = m(x.to(DEV))
    _close(got, want, 2e-4)
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            _close(v.float(), sd["rb." + k].float(), 1e-5)
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            ResidualBlock(32, 64, stride=2).to(DEV)(x[:, :32].to(DEV))


@pytest.mark.parametrize("cin,cout,training,hw", [(32, 64, True, (8, 12)), (64, 64, True, (7, 9)), (64, 128, False, (8, 12)),
                                                   (128, 128, True, (6, 10)), (128, 256, True, (5, 8)),
                                                   (96, 192, True, (6, 9)), (32, 256, True, (4, 6)),
                                                   (160, 64, False, (5, 5)), (256, 256, True, (3, 4)),
                                                   (192, 192, True, (2, 2)), (64, 64, True, (33, 70))])
def test_residual_block_alone_backward(cin, cout, training, hw):
    """loss.backward() through a ResidualBlock used on its own (the reference's blocks are ordinary differentiable
    modules, multimodal_model.py:45-70): gradient of a random linear functional of the output with respect to the input
    and to every parameter, against autograd through the oracle in double precision.  Projection and identity
    shortcut, batch and running statistics, even and odd map sizes."""
    from mmsa_amd.model import ResidualBlock
    torch.manual_seed(cin * 3 + cout)
    m = ResidualBlock(cin, cout).to(DEV).train(training)
    with torch.no_grad():
        for k, p in m.named_parameters():
            if "se.fc" in k:
                p.normal_(0, 0.3)
            elif "bn" in k or k.startswith("shortcut.1"):
                p.normal_(1.0 if k.endswith("weight") else 0.0, 0.3)
        for k, b in m.named_buffers():
            if k.endswith("running_mean"):
                b.normal_(0, 0.2)
            elif k.endswith("running_var"):
                b.uniform_(0.5, 1.5)
    H, W = hw
    x = torch.randn(4, cin, H, W)
    R = torch.randn(4, cout, H, W)
    sd = {k: v.double() if v.is_floating_point() else v for k, v in _sd(m, "rb").items()}
    sd = model_ref.clone_state(sd, requires_grad=True)
    xr = x.double().requires_grad_(True)
    (model_ref.residual_block(sd, "rb", xr, training) * R.double()).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    out = m(xg)
    (out * R.to(DEV)).sum().backward()
    _grad_close(xg.grad, xr.grad, 5e-4)
    for k, p in m.named_parameters():
        _grad_close(p.grad, sd["rb." + k].grad, 5e-4)


def test_normalizer_and_fusion_alone(monkeypatch):
    from mmsa_amd.model import ModalityFusion, MultifactorFeatureNormalizer
    torch.manual_seed(5)
    fn = MultifactorFeatureNormalizer(256, 384).to(DEV)
    fu = ModalityFusion(256, 384, hidden_dim=256, dropout_rate=0.0).to(DEV)
    with torch.no_grad():
        fn.audio_scale.fill_(1.3)
        fn.text_scale.fill_(0.7)
        for p in list(fn.audio_norm.parameters()) + list(fn.text_norm.parameters()):
            p.normal_(0.5, 0.5)
    a, t = torch.randn(6, 256), torch.randn(6, 384) * 2 + 1
    sd = {**_sd(fn, "feature_normalizer"), **_sd(fu, "fusion_layer")}
    with torch.no_grad():
        an, tn = fn(a.to(DEV), t.to(DEV))
    _close(an, model_ref._ln(sd, "feature_normalizer.audio_norm", a) * 1.3)
    _close(tn, model_ref._ln(sd, "feature_normalizer.text_norm", t) * 0.7)
    fu.eval()
    with torch.no_grad():
        _close(fu(an, tn), model_ref.fusion_features(sd, a, t, False, False))
    # training mode: the two host draws decide which modality is zeroed (audio first), as in the reference
    fu.train()
    for da, dt in ((True, False), (False, True), (True, True)):
        monkeypatch.setattr(fu, "draw_modality_dropout", lambda d=(da, dt): d)
        with torch.no_grad():
            _close(fu(an, tn), model_ref.fusion_features(sd, a, t, da, dt))
    monkeypatch.undo()
    torch.manual_seed(0)
    draws = [torch.rand(1).item() < 0.3, torch.rand(1).item() < 0.3]
    torch.manual_seed(0)
    with torch.no_grad():
        _close(fu(an, tn), model_ref.fusion_features(sd, a, t, draws[0], draws[1]))


def test_normalizer_and_fusion_alone_backward(monkeypatch):
    """loss.backward() through the feature normaliser and the fusion layer on their own (multimodal_model.py:216-225,
    263-285), with a dropped modality included: gradients against autograd through the oracle."""
    from mmsa_amd.model import ModalityFusion, MultifactorFeatureNormalizer
    torch.manual_seed(6)
    fn = MultifactorFeatureNormalizer(256, 384).to(DEV)
    fu = ModalityFusion(256, 384, hidden_dim=256, dropout_rate=0.0).to(DEV).train()
    with torch.no_grad():
        fn.audio_scale.fill_(1.3)
        fn.text_scale.fill_(0.7)
        for p in list(fn.audio_norm.parameters()) + list(fn.text_norm.parameters()):
            p.normal_(0.5, 0.5)
    a, t = torch.randn(6, 256), torch.randn(6, 384) * 2 + 1
    R = torch.randn(6, 256)
    for da, dt in ((False, False), (True, False), (False, True)):
        monkeypatch.setattr(fu, "draw_modality_dropout", lambda d=(da, dt): d)
        sd = model_ref.clone_state({**_sd(fn, "feature_normalizer"), **_sd(fu, "fusion_layer")}, requires_grad=True)
        ar, tr = a.clone().requires_grad_(True), t.clone().requires_grad_(True)
        (model_ref.fusion_features(sd, ar, tr, da, dt) * R).sum().backward()
        for p in list(fn.parameters()) + list(fu.parameters()):
            p.grad = None
        ag, tg = a.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
        an, tn = fn(ag, tg)
        (fu(an, tn) * R.to(DEV)).sum().backward()
        _grad_close(ag.grad, ar.grad if ar.grad is not None else torch.zeros_like(a))    # dropped: no gradient
        _grad_close(tg.grad, tr.grad if tr.grad is not None else torch.zeros_like(t))
        for mod, prefix in ((fn, "feature_normalizer"), (fu, "fusion_layer")):
            for k, p in mod.named_parameters():
                want = sd[f"{prefix}.{k}"].grad
                _grad_close(p.grad, want if want is not None else torch.zeros_like(p).cpu())


def test_attention_alone_backward():
    from mmsa_amd.model import Attention
    torch.manual_seed(3)
    m = Attention(64).to(DEV)
    x = torch.randn(2, 5, 64)
    R = torch.randn(2, 5, 64)
    ref = {n: (getattr(m, n).weight.detach().cpu().clone().requires_grad_(True),
               getattr(m, n).bias.detach().cpu().clone().requires_grad_(True)) for n in ("query", "key", "value")}
    xr = x.clone().requires_grad_(True)
    q, k, v = (torch.nn.functional.linear(xr, *ref[n]) for n in ("query", "key", "value"))
    ((torch.softmax(q @ k.transpose(-2, -1) / 8.0, dim=-1) @ v) * R).sum().backward()
    xg = x.to(DEV).requires_grad_(True)
    (m(xg) * R.to(DEV)).sum().backward()
    _grad_close(xg.grad, xr.grad)
    for n in ("query", "key", "value"):
        _grad_close(getattr(m, n).weight.grad, ref[n][0].grad)
        _grad_close(getattr(m, n).bias.grad, ref[n][1].grad)


def test_attention_alone():
    from mmsa_amd.model import Attention
    torch.manual_seed(2)
    m = Attention(64).to(DEV)
    x = torch.randn(2, 5, 64)
    q, k, v = (torch.nn.functional.linear(x, getattr(m, n).weight.cpu(), getattr(m, n).bias.cpu())
               for n in ("query", "key", "value"))
    want = torch.softmax(q @ k.transpose(-2, -1) / 8.0, dim=-1) @ v
    with torch.no_grad():
        _close(m(x.to(DEV)), want.detach())

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
    with pytest.raises(RuntimeError, match="no_grad"):
        m(x.to(DEV))                       # parameters require grad and autograd is on


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

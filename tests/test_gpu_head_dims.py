"""The one-launch fusion head at dimensions other than the reference model's (`CNN implementation/multimodal_model.py:
216-225, 263-285, 393-404` with other widths): the staged kernels (dimensions up to the reference's), the plain kernels
(any dimensions up to 512) and stock torch layers on the CPU must agree on the logits and on every gradient."""
from types import SimpleNamespace

import pytest
import torch
import torch.nn as nn

from conftest import SMALL_VOCAB  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(A, T, Hd, N0, N1, C):
    from mmsa_amd.model import ModalityFusion, MultifactorFeatureNormalizer
    fn = MultifactorFeatureNormalizer(A, T)
    fu = ModalityFusion(A, T, hidden_dim=Hd, dropout_rate=0.0)
    cl = nn.Sequential(nn.Linear(Hd, N0), nn.LayerNorm(N0), nn.ReLU(), nn.Dropout(0.0),
                       nn.Linear(N0, N1), nn.LayerNorm(N1), nn.ReLU(), nn.Dropout(0.0), nn.Linear(N1, C))
    with torch.no_grad():
        fn.audio_scale.fill_(1.2)
        fn.text_scale.fill_(0.8)
        for m in list(fn.modules()) + list(fu.modules()) + list(cl.modules()):
            if isinstance(m, nn.LayerNorm):
                m.weight.normal_(1.0, 0.3)
                m.bias.normal_(0.0, 0.3)
    return SimpleNamespace(feature_normalizer=fn, fusion_layer=fu, classifier=cl)


def _reference(model, a, t, labels):
    fn, fu, cl = model.feature_normalizer, model.fusion_layer, model.classifier
    an = fn.audio_norm(a) * fn.audio_scale
    tn = fn.text_norm(t) * fn.text_scale
    pa = torch.relu(fu.audio_projector[1](fu.audio_projector[0](an)))
    pt = torch.relu(fu.text_projector[1](fu.text_projector[0](tn)))
    w = torch.softmax(fu.attention[0](torch.cat([pa, pt], 1)), 1)
    logits = cl(pa * w[:, :1] + pt * w[:, 1:])
    return logits, torch.nn.functional.cross_entropy(logits, labels)


def _run(model, a, t, labels):
    from mmsa_amd import head
    from mmsa_amd.workspace import Grads, Workspace
    ws = Workspace(DEV, persistent=False)
    grads = Grads()
    st = head.head_forward_both(model, a, t, labels, True, None, None, 0.0, 0, True, ws)
    d_a, d_t = head.head_backward_both(model, st, st.dlogits, grads, ws)
    for fn_ in st.side_work:
        fn_()
    torch.cuda.synchronize()
    return st.logits.clone(), d_a.clone(), d_t.clone(), grads


@pytest.mark.parametrize("dims", [(100, 52, 68, 36, 12, 5), (256, 384, 256, 128, 64, 3), (64, 384, 128, 128, 64, 7),
                                  (260, 100, 64, 64, 32, 2), (32, 32, 256, 128, 4, 3), (512, 512, 512, 256, 128, 10),
                                  (300, 200, 320, 132, 68, 4)])
def test_one_launch_head_at_other_dimensions(monkeypatch, dims):
    A, T, Hd, N0, N1, C = dims
    B = 5
    torch.manual_seed(sum(dims))
    cpu = _build(*dims)
    a, t = torch.randn(B, A), torch.randn(B, T) * 1.5 + 0.3
    labels = torch.randint(0, C, (B,))
    ar, tr = a.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref_logits, ref_loss = _reference(cpu, ar, tr, labels)
    ref_loss.backward()
    import copy
    gpu = SimpleNamespace(**{k: copy.deepcopy(v).to(DEV) for k, v in vars(cpu).items()})
    monkeypatch.setenv("MMSA_HEAD", "")
    results = {}
    for staged in ("1", "0"):
        monkeypatch.setenv("MMSA_HEAD_STAGED", staged)
        results[staged] = _run(gpu, a.to(DEV), t.to(DEV), labels.to(DEV))

    def close(got, want, tol=2e-4):
        want = want.detach()
        err = float((got.detach().cpu() - want).abs().max())
        assert err <= tol * max(1e-3, float(want.abs().max())) + 1e-6, (err, float(want.abs().max()))

    for staged, (logits, d_a, d_t, grads) in results.items():
        close(logits, ref_logits)
        close(d_a, ar.grad)
        close(d_t, tr.grad)
        for part in ("feature_normalizer", "fusion_layer", "classifier"):
            for (k, pc), (_, pg) in zip(getattr(cpu, part).named_parameters(), getattr(gpu, part).named_parameters()):
                close(grads.of(pg), pc.grad)
    # the two kernel families run the same arithmetic
    l1, l0 = results["1"][0], results["0"][0]
    assert float((l1 - l0).abs().max()) <= 2e-6 * max(1.0, float(l0.abs().max()))

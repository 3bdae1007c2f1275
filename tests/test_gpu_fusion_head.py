"""Attention-fusion head (config 4: text [8,128] + audio [8,768] pooled features, fusion_dim 512, grad-accum 2)
against the CPU restatement in oracle/fusion_head_ref.py (parity unpinned: see that file's header)."""
import numpy as np
import pytest
import torch

from oracle import fusion_head_ref

DEV = "cuda"
EXPECTED_KEYS = ["fusion_layer.text_projection.weight", "fusion_layer.text_projection.bias",
                 "fusion_layer.audio_projection.weight", "fusion_layer.audio_projection.bias",
                 "fusion_layer.text_norm.weight", "fusion_layer.text_norm.bias", "fusion_layer.audio_norm.weight",
                 "fusion_layer.audio_norm.bias", "fusion_layer.attention_layer.0.weight",
                 "fusion_layer.attention_layer.0.bias", "fusion_layer.attention_layer.3.weight",
                 "fusion_layer.attention_layer.3.bias", "fusion_layer.gate.0.weight", "fusion_layer.gate.0.bias",
                 "classifier.0.weight", "classifier.0.bias", "classifier.1.weight", "classifier.1.bias",
                 "classifier.4.weight", "classifier.4.bias", "classifier.5.weight", "classifier.5.bias",
                 "classifier.8.weight", "classifier.8.bias"]


# The two attention scores pass through a softmax, which ignores a common shift: the bias of the last score layer
# has an exactly-zero true gradient and every implementation returns rounding noise for it (Adam then amplifies it).
SHIFT_BIAS = "fusion_layer.attention_layer.3.bias"


def make(dropout=0.0):
    from mmsa_amd.fusion_head import HybridFusionHead
    torch.manual_seed(42)
    return HybridFusionHead(128, 768, 512, 3, dropout)


def feats(seed, B=8):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 128, generator=g), torch.randn(B, 768, generator=g), torch.randint(0, 3, (B,), generator=g)


def test_state_dict_keys_and_init():
    m = make()
    sd = m.state_dict()
    assert list(sd.keys()) == EXPECTED_KEYS
    assert sd["fusion_layer.gate.0.weight"].shape == (512, 1024)
    assert float(sd["classifier.0.bias"].abs().max()) == 0.0
    std = float(sd["fusion_layer.text_projection.weight"].std())
    assert abs(std - 0.5 * (2.0 / (128 + 512)) ** 0.5) < 0.1 * std          # xavier normal, gain 0.5


@pytest.mark.gpu
def test_forward_backward_match_oracle():
    m = make().to(DEV).train()
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    t, a, y = feats(1)
    out = m(t.to(DEV), a.to(DEV), labels=y.to(DEV))
    out["loss"].backward()
    w = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = fusion_head_ref.forward(w, t.double(), a.double(), y)
    ref["loss"].backward()
    np.testing.assert_allclose(out["logits"].detach().cpu().numpy(), ref["logits"].detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(out["attention_weights"].detach().cpu().numpy(), ref["attention_weights"].detach().numpy(),
                               rtol=1e-4, atol=1e-6)
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-5
    assert torch.equal(out["logits"].argmax(1).cpu(), ref["logits"].argmax(1))
    for k, p in m.named_parameters():
        g, r = p.grad.detach().cpu().double(), w[k].grad
        if k == SHIFT_BIAS:
            assert float(g.abs().max()) < 1e-6
            continue
        err = float(torch.linalg.vector_norm(g - r) / (torch.linalg.vector_norm(r) + 1e-30))
        assert err < 1e-4, f"{k}: relative L2 error {err:.2e}"
    # float -1/0/1 labels are shifted by one, like the reference
    out2 = m(t.to(DEV), a.to(DEV), labels=(y.float() - 1).to(DEV))
    assert abs(float(out2["loss"].detach()) - float(ref["loss"].detach())) < 1e-5


@pytest.mark.gpu
def test_grad_accum_training_matches_oracle():
    """Six micro-batches of 8 with accumulate 2, clip 1.0, AdamW(1e-4, wd 0.005), cosine warm-up per optimiser step."""
    from transformers import get_cosine_schedule_with_warmup
    m = make().to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ref_p = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}

    def run(params, fwd, dev):
        opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.005)
        sch = get_cosine_schedule_with_warmup(opt, 1, 3)
        losses = []
        opt.zero_grad()
        for s in range(6):
            t, a, y = feats(10 + s)
            loss = fwd(t.to(dev), a.to(dev), y.to(dev)) / 2
            loss.backward()
            losses.append(float(loss.detach()) * 2)
            if (s + 1) % 2 == 0:
                torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)
                opt.step()
                sch.step()
                opt.zero_grad()
        return losses

    la = run(list(m.parameters()), lambda t, a, y: m(t, a, labels=y)["loss"], DEV)
    lb = run(list(ref_p.values()), lambda t, a, y: fusion_head_ref.forward(ref_p, t, a, y)["loss"], "cpu")
    np.testing.assert_allclose(la, lb, rtol=0, atol=2e-5)
    for k, p in m.named_parameters():
        if k == SHIFT_BIAS:
            continue
        upd = float(torch.linalg.vector_norm(ref_p[k].detach() - sd0[k])) + 1e-12
        err = float(torch.linalg.vector_norm(p.detach().cpu() - ref_p[k].detach()))
        assert err < 0.02 * upd + 1e-7, k


@pytest.mark.gpu
def test_eval_and_layer_alone():
    m = make(dropout=0.3).to(DEV).eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    t, a, _ = feats(3, B=5)
    with torch.no_grad():
        out = m(t.to(DEV), a.to(DEV))
        fused, w = m.fusion_layer(t.to(DEV), a.to(DEV))
    ref = fusion_head_ref.forward(sd, t, a)
    assert out["loss"] is None
    np.testing.assert_allclose(out["probabilities"].cpu().numpy(), ref["probabilities"].numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(fused.cpu().numpy(), ref["fused"].numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(w.cpu().numpy(), ref["attention_weights"].numpy(), rtol=1e-4, atol=1e-6)
    m.train()                      # dropout active: runs, differs from eval, stays finite
    o2 = m(t.to(DEV), a.to(DEV))
    assert torch.isfinite(o2["logits"]).all()
    # a class label outside the classes: the reference's CrossEntropyLoss raises, so does the head (float labels are
    # sentiments -1 / 0 / 1 and are shifted first, early_fusion.py:341-345)
    for bad in (torch.tensor([0, 1, 3, 2, 0]), torch.tensor([0, -1, 1, 2, 0]), torch.tensor([0., 2., 1., -1., 0.])):
        with pytest.raises(IndexError, match="out of bounds"):
            m(t.to(DEV), a.to(DEV), labels=bad.to(DEV))
    assert m(t.to(DEV), a.to(DEV), labels=torch.tensor([-1., 0., 1., 1., 0.]).to(DEV))["loss"] is not None


@pytest.mark.gpu
@pytest.mark.parametrize("td,ad,fd,nc,B", [(128, 768, 512, 3, 1), (128, 768, 512, 3, 33), (64, 256, 128, 2, 8),
                                           (100, 300, 96, 5, 7), (768, 768, 1024, 3, 4), (32, 40, 64, 7, 130)])
def test_other_widths_and_batch_sizes_match_oracle(td, ad, fd, nc, B):
    """Feature, fusion and class widths other than the reference's, including odd ones, and the gradients that flow
    back into the pooled features (the encoders upstream of this head receive them)."""
    from mmsa_amd.fusion_head import HybridFusionHead
    torch.manual_seed(7)
    m = HybridFusionHead(td, ad, fd, nc, 0.0).to(DEV).train()
    with torch.no_grad():                      # the reference initialises biases to zero: make them matter here
        for k, p in m.named_parameters():
            if k.endswith(".bias"):
                p.normal_(0, 0.1)
    sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    t, a, y = torch.randn(B, td, generator=g), torch.randn(B, ad, generator=g), torch.randint(0, nc, (B,), generator=g)
    tg, ag = t.to(DEV).requires_grad_(True), a.to(DEV).requires_grad_(True)
    out = m(tg, ag, labels=y.to(DEV))
    out["loss"].backward()
    w = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    tr, ar = t.double().requires_grad_(True), a.double().requires_grad_(True)
    ref = fusion_head_ref.forward(w, tr, ar, y)
    ref["loss"].backward()
    np.testing.assert_allclose(out["logits"].detach().cpu().numpy(), ref["logits"].detach().numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(out["attention_weights"].detach().cpu().numpy(), ref["attention_weights"].detach().numpy(),
                               rtol=2e-4, atol=1e-6)
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 2e-5
    pairs = [(k, p.grad, w[k].grad) for k, p in m.named_parameters()] + [("text", tg.grad, tr.grad), ("audio", ag.grad, ar.grad)]
    for k, gg, r in pairs:
        gg = gg.detach().cpu().double()
        if k == SHIFT_BIAS:
            assert float(gg.abs().max()) < 1e-6
            continue
        err = float(torch.linalg.vector_norm(gg - r) / (torch.linalg.vector_norm(r) + 1e-30))
        assert err < 2e-4, f"{k}: relative L2 error {err:.2e}"

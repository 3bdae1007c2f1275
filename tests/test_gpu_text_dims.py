"""TextModel (`CNN implementation/multimodal_model.py:140-199`) at configurations other than the reference's default
(embedding width, filter widths and counts, sequence length): features and every gradient against stock torch
functions on the CPU, with repeated tokens and padding in the batch."""
import pytest
import torch
import torch.nn.functional as F

from conftest import SMALL_VOCAB  # noqa: F401  (puts the package on sys.path)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _reference(weights, ids):
    emb = F.embedding(ids, weights["embedding.weight"], padding_idx=0).transpose(1, 2)
    feats = []
    i = 0
    while f"convs.{i}.weight" in weights:
        y = F.relu(F.conv1d(emb, weights[f"convs.{i}.weight"], weights[f"convs.{i}.bias"]))
        feats.append(y.max(dim=2).values)
        i += 1
    return torch.cat(feats, dim=1)


CONFIGS = [(128, [2, 3], 64, 40, 6), (64, [3, 4, 5, 6, 7], 128, 33, 5), (256, [7], 192, 16, 3), (32, [3, 4, 5], 64, 128, 9),
           (96, [2, 5, 7, 3], 64, 21, 4), (64, [2, 5, 7, 3], 64, 21, 4), (192, [3, 4, 5], 128, 128, 32),
           (256, [3, 4, 5], 128, 300, 4), (256, [3, 4, 5], 128, 1000, 2)]      # sequences longer than the reference's 128


@pytest.mark.parametrize("E,sizes,nf,L,B", CONFIGS)
def test_text_model_at_other_configurations(E, sizes, nf, L, B):
    from mmsa_amd.model import TextModel
    torch.manual_seed(E + nf + L)
    V = 300
    m = TextModel(vocab_size=V, embedding_dim=E, filter_sizes=sizes, num_filters=nf, max_length=L).to(DEV)
    with torch.no_grad():
        for c in m.convs:
            c.bias.normal_(0, 0.1)
    ids = torch.randint(1, V, (B, L))
    ids[:, L // 2:] = ids[:, : L - L // 2]          # every token of the first half occurs again
    ids[0, -3:] = 0                                  # padding
    if B > 1:
        ids[1] = 7                                   # one token all along a row
    R = torch.randn(B, len(sizes) * nf)
    ref = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    want = _reference(ref, ids)
    (want * R).sum().backward()

    def close(g, w, tol=2e-4):
        w = w.detach()
        err = float((g.detach().cpu() - w).abs().max())
        assert err <= tol * max(1e-3, float(w.abs().max())) + 1e-6, (err, float(w.abs().max()))

    with torch.no_grad():
        close(m(ids.to(DEV)), want)
    if len(sizes) > 4 or E % 64:
        # the forward takes these shapes; the backward kernels do not, and say so instead of failing in a launcher
        with pytest.raises(NotImplementedError, match="backward"):
            m(ids.to(DEV))
        return
    got = m(ids.to(DEV))
    (got * R.to(DEV)).sum().backward()
    close(got, want)
    for k, p in m.named_parameters():
        close(p.grad, ref[k].grad)
    assert float(m.embedding.weight.grad[0].abs().max()) == 0.0   # the padding row

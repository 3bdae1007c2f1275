import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-sentiment-analysis_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SMALL_VOCAB = 1000


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    out = {}
    for name in ("init_fingerprint", "small_cases", "loss_curves", "focal"):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            out[name] = json.load(f)
    out["npz"] = dict(np.load(os.path.join(GOLDEN, "small_cases.npz")))
    return out


def seeded_model(modality, vocab=None, dropout0=True, **kw):
    """Seed-42 construction of the drop-in model (its initial weights equal the reference's, which
    tests/test_oracle_golden.py::test_seeded_init_matches_reference proves against fingerprints captured from the
    reference)."""
    from mmsa_amd.model import MultimodalSentimentModel
    torch.manual_seed(42)
    args = dict(modality=modality)
    if dropout0:
        args.update(audio_dropout=0.0, text_dropout=0.0, fusion_dropout=0.0)
    if vocab is not None:
        args["vocab_size"] = vocab
    args.update(kw)
    return MultimodalSentimentModel(**args)


def batch_tensors(start, n, vocab=30522, device=None, wave=False):
    from mmsa_amd import data
    b = data.make_batch(start, n, vocab_size=vocab, with_wave=wave)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    if device is not None:
        t = {k: v.to(device) for k, v in t.items()}
    return t


def model_inputs(modality, b):
    kw = {}
    if modality in ("audio", "both"):
        kw["audio"] = b["audio"]
    if modality in ("text", "both"):
        kw["input_ids"] = b["input_ids"]
    return kw


def fingerprint_close(t, ref, rtol=1e-4, atol=1e-6):
    """Compare a tensor with a stored fingerprint (shape, sum, L2, leading and strided samples)."""
    f = t.detach().double().flatten().cpu()
    assert list(t.shape) == ref["shape"]
    scale = max(ref["l2"], 1e-12)
    n = f.numel()
    stride = max(n // 16, 1)
    got8 = f[:8].numpy()
    got16 = f[::stride][:16].numpy()
    tol = rtol * scale / max(np.sqrt(n), 1.0) * 50 + atol
    np.testing.assert_allclose(got8, np.array(ref["first8"]), rtol=rtol * 50, atol=tol)
    np.testing.assert_allclose(got16, np.array(ref["strided16"]), rtol=rtol * 50, atol=tol)
    assert abs(float(torch.linalg.vector_norm(f)) - ref["l2"]) <= rtol * scale + atol
    assert abs(float(f.sum()) - ref["sum"]) <= rtol * max(ref["abs_sum"], 1e-12) + atol

"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32, functional) of the CNN multimodal model.

Nothing under ``oracle/`` is part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, as the checker.  The product path is the HIP library.

The restatement works directly on a ``state_dict`` (a flat ``{key: tensor}`` mapping with the
reference's keys) instead of ``nn.Module`` objects, so it shares no structure with the product's
modules.  It is pinned against vectors captured from the reference module itself
(``tests/golden/make_golden.py`` imports ``CNN implementation/multimodal_model.py`` in the build
container and stores inputs/outputs under ``tests/golden/``).

Reference lines each function follows are cited in its docstring; all paths are relative to
``CNN implementation/``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5
L2_LAMBDA = 0.001


def _bn(sd: SD, key: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """BatchNorm2d/1d with torch defaults (eps 1e-5, momentum 0.1); updates running stats in ``sd``
    in place when training, exactly as the stock module does (multimodal_model.py:49,51,58,80,369,373)."""
    rm, rv = sd[key + ".running_mean"], sd[key + ".running_var"]
    out = F.batch_norm(x, rm, rv, sd[key + ".weight"], sd[key + ".bias"], training, BN_MOMENTUM, BN_EPS)
    if training and (key + ".num_batches_tracked") in sd:
        sd[key + ".num_batches_tracked"] += 1
    return out


def se_block(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    """multimodal_model.py:38-42 -- global average -> FC -> ReLU -> FC -> sigmoid -> channel scale."""
    b, c = x.shape[0], x.shape[1]
    y = x.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[key + ".fc.0.weight"]))
    y = torch.sigmoid(F.linear(y, sd[key + ".fc.2.weight"]))
    return x * y.view(b, c, 1, 1)


def residual_block(sd: SD, key: str, x: torch.Tensor, training: bool) -> torch.Tensor:
    """multimodal_model.py:64-70 (stride 1, projection shortcut since channels always change)."""
    out = F.conv2d(x, sd[key + ".conv1.weight"], None, 1, 1)
    out = F.relu(_bn(sd, key + ".bn1", out, training))
    out = F.conv2d(out, sd[key + ".conv2.weight"], None, 1, 1)
    out = _bn(sd, key + ".bn2", out, training)
    out = se_block(sd, key + ".se", out)
    if (key + ".shortcut.0.weight") in sd:
        sc = F.conv2d(x, sd[key + ".shortcut.0.weight"], None, 1, 0)
        sc = _bn(sd, key + ".shortcut.1", sc, training)
    else:
        sc = x
    return F.relu(out + sc)


def audio_features(sd: SD, spec: torch.Tensor, training: bool, key: str = "audio_model") -> torch.Tensor:
    """multimodal_model.py:118-129 -- ``[B,1,64,130] -> [B,256]``."""
    x = F.conv2d(spec, sd[key + ".conv1.0.weight"], sd[key + ".conv1.0.bias"], 1, 1)
    x = F.max_pool2d(F.relu(_bn(sd, key + ".conv1.1", x, training)), 2, 2)
    for i in (1, 2, 3):
        x = F.max_pool2d(residual_block(sd, f"{key}.res_block{i}", x, training), 2, 2)
    return x.mean(dim=(2, 3))


def text_features(sd: SD, input_ids: torch.Tensor, key: str = "text_model") -> torch.Tensor:
    """multimodal_model.py:173-199 -- gather, transpose, per-width conv1d + ReLU + max over time, concat.
    ``attention_mask`` plays no part in the reference's arithmetic."""
    emb = F.embedding(input_ids, sd[key + ".embedding.weight"], padding_idx=0).transpose(1, 2)
    feats = []
    i = 0
    while f"{key}.convs.{i}.weight" in sd:
        c = F.relu(F.conv1d(emb, sd[f"{key}.convs.{i}.weight"], sd[f"{key}.convs.{i}.bias"]))
        feats.append(c.amax(dim=2))
        i += 1
    return torch.cat(feats, dim=1)


def _ln(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], LN_EPS)


def _lin(sd: SD, key: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def fusion_features(sd: SD, a: torch.Tensor, t: torch.Tensor, drop_audio: bool, drop_text: bool,
                    p_drop: float = 0.0, training: bool = False) -> torch.Tensor:
    """multimodal_model.py:216-225 (normaliser) then :263-285 (fusion).  The two host-side modality
    dropout decisions (``torch.rand(1).item() < 0.3`` each, :264-268) are passed in as booleans."""
    a = _ln(sd, "feature_normalizer.audio_norm", a) * sd["feature_normalizer.audio_scale"]
    t = _ln(sd, "feature_normalizer.text_norm", t) * sd["feature_normalizer.text_scale"]
    if drop_audio:
        a = torch.zeros_like(a)
    if drop_text:
        t = torch.zeros_like(t)
    pa = F.relu(_ln(sd, "fusion_layer.audio_projector.1", _lin(sd, "fusion_layer.audio_projector.0", a)))
    pt = F.relu(_ln(sd, "fusion_layer.text_projector.1", _lin(sd, "fusion_layer.text_projector.0", t)))
    pa = F.dropout(pa, p_drop, training)
    pt = F.dropout(pt, p_drop, training)
    w = torch.softmax(_lin(sd, "fusion_layer.attention.0", torch.cat([pa, pt], dim=1)), dim=1)
    return pa * w[:, 0:1] + pt * w[:, 1:2]


def classify(sd: SD, modality: str, feats: torch.Tensor, training: bool, p_drop: float = 0.0) -> torch.Tensor:
    """The three classifier stacks, multimodal_model.py:365-377 (audio), :379-391 (text), :393-404 (both)."""
    x = feats
    if modality == "audio":
        x = F.dropout(F.relu(_bn(sd, "classifier.2", _lin(sd, "classifier.1", x), training)), p_drop, training)
        x = F.dropout(F.relu(_bn(sd, "classifier.6", _lin(sd, "classifier.5", x), training)), p_drop / 2, training)
        return _lin(sd, "classifier.9", x)
    if modality == "text":
        x = F.dropout(x, p_drop, training)
        x = F.dropout(_ln(sd, "classifier.3", F.relu(_lin(sd, "classifier.1", x))), p_drop, training)
        x = F.dropout(_ln(sd, "classifier.7", F.relu(_lin(sd, "classifier.5", x))), p_drop, training)
        return _lin(sd, "classifier.9", x)
    x = F.dropout(F.relu(_ln(sd, "classifier.1", _lin(sd, "classifier.0", x))), p_drop, training)
    x = F.dropout(F.relu(_ln(sd, "classifier.5", _lin(sd, "classifier.4", x))), p_drop / 2, training)
    return _lin(sd, "classifier.8", x)


def param_keys(sd: SD):
    """Keys of trainable tensors in registration order (buffers excluded)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def forward(sd: SD, modality: str, audio: Optional[torch.Tensor] = None,
            input_ids: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None,
            training: bool = False, drop_audio: bool = False, drop_text: bool = False,
            dropouts=(0.0, 0.0, 0.0)):
    """multimodal_model.py:406-464.  Returns ``{"loss","logits","probabilities"}``.

    ``dropouts`` = (audio_dropout, text_dropout, fusion_dropout) ctor values; only nonzero in
    RNG-independent tests at 0.  The L2 term (:456-462) sums ``torch.norm`` of every parameter in
    registration order and is added whenever labels are given in 'both' mode (eval included)."""
    if modality == "audio":
        if audio is None:
            raise ValueError("Audio input is required for audio modality")
        feats = audio_features(sd, audio, training)
        logits = classify(sd, modality, feats, training, dropouts[0])
    elif modality == "text":
        if input_ids is None:
            raise ValueError("Text input is required for text modality")
        feats = text_features(sd, input_ids)
        logits = classify(sd, modality, feats, training, dropouts[1])
    else:
        if audio is None or input_ids is None:
            raise ValueError("Both audio and text inputs are required for multimodal fusion")
        a = audio_features(sd, audio, training)
        t = text_features(sd, input_ids)
        feats = fusion_features(sd, a, t, training and drop_audio, training and drop_text, dropouts[2], training)
        logits = classify(sd, modality, feats, training, dropouts[2])
    probs = torch.softmax(logits, dim=1)
    loss = None
    if labels is not None:
        loss = F.cross_entropy(logits, labels)
        if modality == "both":
            reg = 0
            for k in param_keys(sd):
                reg = reg + torch.norm(sd[k])
            loss = loss + L2_LAMBDA * reg
    return {"loss": loss, "logits": logits, "probabilities": probs}


def clone_state(sd: SD, requires_grad: bool = False) -> SD:
    out = {}
    pk = set(param_keys(sd))
    for k, v in sd.items():
        c = v.detach().clone()
        if requires_grad and k in pk:
            c.requires_grad_(True)
        out[k] = c
    return out


def forward_backward(sd: SD, modality: str, audio, input_ids, labels, drop_audio=False, drop_text=False):
    """Train-mode forward + backward on a private copy of ``sd``; returns (outputs, grads, new_buffers)."""
    w = clone_state(sd, requires_grad=True)
    out = forward(w, modality, audio, input_ids, labels, training=True, drop_audio=drop_audio, drop_text=drop_text)
    out["loss"].backward()
    grads = {k: (w[k].grad.detach().clone() if w[k].grad is not None else torch.zeros_like(w[k]))
             for k in param_keys(w)}
    bufs = {k: v.detach().clone() for k, v in w.items() if k not in grads}
    return {k: (v.detach() if v is not None else None) for k, v in out.items()}, grads, bufs


def focal_loss(logits: torch.Tensor, targets: torch.Tensor, alpha=None, gamma: float = 2.0,
               reduction: str = "mean") -> torch.Tensor:
    """multimodal_model.py:476-480 -- ``(1 - exp(-ce))**gamma * ce`` on per-sample (weighted) CE;
    any reduction other than 'mean' sums."""
    ce = F.cross_entropy(logits, targets, weight=alpha, reduction="none")
    fl = (1.0 - torch.exp(-ce)) ** gamma * ce
    return fl.mean() if reduction == "mean" else fl.sum()

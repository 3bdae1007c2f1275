"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the attention-fusion head (the reference's `Pretrained model implementations/early_fusion.py` head).

PARITY UNPINNED beyond torch.nn itself: the reference module (`Pretrained model implementations/early_fusion.py`)
cannot be imported here (its imports need librosa / torchaudio / seaborn and remote HuggingFace weights) and ships no
vectors.  This file restates, from the source text, `AttentionFusionLayer.forward` (`early_fusion.py:127-158`), the
classifier stack (`:212-222`) and the loss (`:336-347`) as a function of a state_dict with the reference's keys;
every primitive is a stock torch.nn.functional op, so the restatement is mechanically checkable.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def forward(sd, text, audio, labels=None):
    def lin(k, x):
        return F.linear(x, sd[k + ".weight"], sd[k + ".bias"])

    def ln(k, x):
        return F.layer_norm(x, (x.shape[-1],), sd[k + ".weight"], sd[k + ".bias"], 1e-5)

    tp = ln("fusion_layer.text_norm", lin("fusion_layer.text_projection", text))
    ap = ln("fusion_layer.audio_norm", lin("fusion_layer.audio_projection", audio))
    mod = torch.stack([tp, ap], dim=1)                                         # [B, 2, Fd]
    sc = lin("fusion_layer.attention_layer.3", torch.tanh(lin("fusion_layer.attention_layer.0", mod)))
    w = torch.softmax(sc.squeeze(-1), dim=1)                                   # [B, 2]
    wt, wa = tp * w[:, 0:1], ap * w[:, 1:2]
    gate = torch.sigmoid(lin("fusion_layer.gate.0", torch.cat([wt, wa], dim=1)))
    fused = wt * gate + wa * (1 - gate)
    x = F.gelu(ln("classifier.1", lin("classifier.0", fused)))
    x = F.gelu(ln("classifier.5", lin("classifier.4", x)))
    logits = lin("classifier.8", x)
    probs = torch.softmax(logits, dim=1)
    loss = None
    if labels is not None:
        if labels.dim() == 1 and labels.dtype != torch.long:
            labels = (labels + 1).long()
        loss = F.cross_entropy(logits, labels)
    return {"loss": loss, "logits": logits, "probabilities": probs, "attention_weights": w, "fused": fused}

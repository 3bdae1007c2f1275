"""Attention-fusion head over pooled backbone features (the reference's `Pretrained model implementations/early_fusion.py` head).

Mirrors `Pretrained model implementations/early_fusion.py`: ``AttentionFusionLayer`` (`:83-158`: per-modality
Linear + LayerNorm, a shared tanh attention MLP scoring the two projections, softmax over the two scores, sigmoid
gate on the concatenation, gated sum) and the classifier of ``HybridMultimodalModel`` (`:212-222`: Linear, LayerNorm,
GELU, Dropout twice, Linear), xavier-normal(gain 0.5) / zero-bias init (`:120-125,279-286`), softmax + mean
cross-entropy (`:336-347`).  The HuggingFace backbones that produce the ``[B,128]`` text and ``[B,768]`` audio
features are outside this build; the head takes the pooled features directly.  Parameter names equal the
reference's (``fusion_layer.*``, ``classifier.{0,1,4,5,8}.*``), so those entries of its checkpoints load.

``loss.backward()`` works (one autograd node spans the head, forward and backward are HIP launch sequences), so the
reference's loop -- loss / grad_accum_steps, backward, clip_grad_norm_ every second step, AdamW(lr 1e-4, wd 0.005),
cosine warm-up scheduler stepped per optimiser step (`:642-650,716-733`) -- runs unchanged on it.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import ops
from .head import _Dense, _finish
from .model import _needs_grad, _require_gpu, _stack
from .workspace import Grads, Workspace


def _xavier_zero(modules):
    for m in modules:
        if isinstance(m, nn.Linear):
            nn.init.xavier_normal_(m.weight, gain=0.5)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


class AttentionFusionLayer(nn.Module):
    def __init__(self, text_dim, audio_dim, fusion_dim=512, dropout_rate=0.3):
        super().__init__()
        self.text_dim, self.audio_dim, self.fusion_dim = text_dim, audio_dim, fusion_dim
        self.dropout_rate = dropout_rate
        self.text_projection = nn.Linear(text_dim, fusion_dim)
        self.audio_projection = nn.Linear(audio_dim, fusion_dim)
        self.text_norm = nn.LayerNorm(fusion_dim)
        self.audio_norm = nn.LayerNorm(fusion_dim)
        self.attention_layer = nn.Sequential(nn.Linear(fusion_dim, fusion_dim // 2), nn.Tanh(),
                                             nn.Dropout(dropout_rate), nn.Linear(fusion_dim // 2, 1))
        self.gate = nn.Sequential(nn.Linear(fusion_dim * 2, fusion_dim), nn.Sigmoid())
        _xavier_zero(self.modules())

    def forward(self, text_features, audio_features):
        """Returns (fused_features [B, fusion_dim], attention_weights [B, 2]) like the reference layer."""
        dev = _require_gpu(self, "AttentionFusionLayer")
        t = text_features.to(device=dev, dtype=torch.float32).contiguous()
        a = audio_features.to(device=dev, dtype=torch.float32).contiguous()
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or t.requires_grad
                                        or a.requires_grad):
            return _FusionFn.apply(self, None, t, a, None, *self.parameters())[:2]
        ws = self.__dict__.setdefault("_ws", Workspace(dev))
        st = fusion_forward(self, t, a, self.training, self._seed(), ws)
        return st.fused.clone(), st.wts.view(-1, 2).clone()

    def _seed(self):
        self.__dict__["_calls"] = self.__dict__.get("_calls", 0) + 1
        return (torch.initial_seed() * 7919 + self._calls * 131) & ((1 << 62) - 1)


def fusion_forward(fl: AttentionFusionLayer, text, audio, training, seed, ws: Workspace):
    B, Fd = text.shape[0], fl.fusion_dim
    p = fl.dropout_rate if training else 0.0
    st = SimpleNamespace(B=B, Fd=Fd, text=text, audio=audio)
    st.M = ws.buf("fh.M", (B, 2 * Fd))
    st.tp = _Dense(ws, "fh.tp", fl.text_projection, fl.text_norm, ops.EPI_LN, B)
    st.ap = _Dense(ws, "fh.ap", fl.audio_projection, fl.audio_norm, ops.EPI_LN, B)
    st.tp.fwd(text, st.M[:, :Fd])
    st.ap.fwd(audio, st.M[:, Fd:])
    st.M2 = st.M.view(2 * B, Fd)                 # rows 2b / 2b+1 = text / audio projection of utterance b
    st.a0 = _Dense(ws, "fh.a0", fl.attention_layer[0], None, ops.EPI_TANH, 2 * B, p, seed + 1)
    st.a3 = _Dense(ws, "fh.a3", fl.attention_layer[3], None, ops.EPI_NONE, 2 * B)
    h = st.a0.fwd(st.M2, ws.buf("fh.att_h", (2 * B, Fd // 2)))
    st.scores = st.a3.fwd(h, ws.buf("fh.scores", (2 * B, 1)))
    st.wts = ws.buf("fh.wts", 2 * B)
    st.Wm = ws.buf("fh.Wm", (B, 2 * Fd))
    ops.pair_attn_mix_fwd(st.M, st.scores, st.wts, st.Wm)
    st.g = _Dense(ws, "fh.gate", fl.gate[0], None, ops.EPI_SIGMOID, B)
    st.gate = st.g.fwd(st.Wm, ws.buf("fh.gate_out", (B, Fd)))
    st.fused = ws.buf("fh.fused", (B, Fd))
    ops.gate_mix_fwd(st.Wm, st.gate, st.fused)
    return st


def fusion_backward(fl, st, dfused, grads: Grads, ws: Workspace, want_dtext=False, want_daudio=False):
    B, Fd = st.B, st.Fd
    dWm = ws.buf("fh.dWm", (B, 2 * Fd))
    dgate = ws.buf("fh.dgate", (B, Fd))
    ops.gate_mix_bwd(dfused, st.Wm, st.gate, dWm, dgate)
    st.g.bwd(ws, dgate, dWm, grads, accumulate_dx=True)
    dM = ws.buf("fh.dM", (B, 2 * Fd))
    dsc = ws.buf("fh.dscores", (2 * B, 1))
    ops.pair_attn_mix_bwd(dWm, st.M, st.wts, dM, dsc)
    dh = ws.buf("fh.dh", (2 * B, Fd // 2))
    st.a3.bwd(ws, dsc, dh, grads)
    st.a0.bwd(ws, dh, dM.view(2 * B, Fd), grads, accumulate_dx=True)
    dt = ws.buf("fh.dtext", st.text.shape) if want_dtext else None
    da = ws.buf("fh.daudio", st.audio.shape) if want_daudio else None
    st.tp.bwd(ws, dM[:, :Fd], dt, grads)
    st.ap.bwd(ws, dM[:, Fd:], da, grads)
    from .audio import fork
    fork(ws, dfused.device, 2).join()
    return dt, da


class _FusionFn(torch.autograd.Function):
    """Autograd node for the fusion layer alone (head=None) or fusion layer + classifier + CE (head given)."""

    @staticmethod
    def forward(ctx, fl, head, text, audio, labels, *params):
        ws = Workspace(text.device, persistent=False)
        training = fl.training
        st = fusion_forward(fl, text, audio, training, fl._seed(), ws)
        ctx.fl, ctx.head, ctx.st, ctx.ws = fl, head, st, ws
        ctx.need_in = (text.requires_grad, audio.requires_grad)
        ctx.set_materialize_grads(False)
        wts = st.wts.view(-1, 2).clone()
        if head is None:
            return st.fused.clone(), wts, None, None, None
        hs = head._classify(st.fused, labels, ws)
        ctx.hs = hs
        loss = hs.loss.view(()).clone() if hs.loss is not None else None
        return st.fused.clone(), wts, hs.logits.clone(), hs.probs.clone(), loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g_fused, g_wts, g_logits, g_probs, g_loss):
        fl, head, st, ws = ctx.fl, ctx.head, ctx.st, ctx.ws
        plist = list(fl.parameters()) + (list(head.classifier.parameters()) if head is not None else [])
        grads = Grads()
        dfused = None
        if head is not None:
            hs = ctx.hs
            d = None
            if g_loss is not None and hs.dlogits is not None:
                d = hs.dlogits * g_loss
            if g_logits is not None:
                d = g_logits if d is None else d + g_logits
            if g_probs is not None:
                p = hs.probs
                dp = p * (g_probs - (g_probs * p).sum(dim=1, keepdim=True))
                d = dp if d is None else d + dp
            if d is not None:
                dfused = head._classify_backward(hs, d.contiguous(), grads, ws)
        if g_fused is not None:
            dfused = g_fused.contiguous() if dfused is None else dfused + g_fused
        if g_wts is not None:
            raise NotImplementedError("gradients through the returned attention weights are not supported")
        if dfused is None:
            return (None,) * 5 + tuple(None for _ in plist)
        dt, da = fusion_backward(fl, st, dfused, grads, ws, *ctx.need_in)
        out = []
        for p in plist:
            g = grads.get(p)
            out.append(g if g is not None else torch.zeros_like(p))
        return (None, None, dt, da, None) + tuple(out)


class HybridFusionHead(nn.Module):
    """``fusion_layer`` + ``classifier`` of the reference's HybridMultimodalModel, fed with pooled features.

    ``forward(text_features [B,text_dim], audio_features [B,audio_dim], labels=None)`` returns the reference's
    dict keys ``loss, logits, probabilities, attention_weights`` (labels in {-1,0,1} float are shifted by one, as
    early_fusion.py:341-345 does)."""

    def __init__(self, text_dim=128, audio_dim=768, fusion_dim=512, num_classes=3, dropout_rate=0.3):
        super().__init__()
        self.dropout_rate = dropout_rate
        self.fusion_layer = AttentionFusionLayer(text_dim, audio_dim, fusion_dim, dropout_rate)
        self.classifier = _stack([("lin", fusion_dim, 256), ("ln", 256), ("gelu",), ("drop", dropout_rate),
                                  ("lin", 256, 64), ("ln", 64), ("gelu",), ("drop", dropout_rate),
                                  ("lin", 64, num_classes)])
        self.loss_fn = nn.CrossEntropyLoss()
        _xavier_zero(self.fusion_layer.modules())
        _xavier_zero(self.classifier.modules())

    def _classify(self, fused, labels, ws):
        cl = self.classifier
        B = fused.shape[0]
        p = self.dropout_rate if self.training else 0.0
        seed = self.fusion_layer._seed()
        hs = SimpleNamespace(B=B)
        hs.c0 = _Dense(ws, "fh.c0", cl[0], cl[1], ops.EPI_LN_GELU, B, p, seed + 11)
        hs.c1 = _Dense(ws, "fh.c1", cl[4], cl[5], ops.EPI_LN_GELU, B, p, seed + 12)
        hs.c2 = _Dense(ws, "fh.c2", cl[8], None, ops.EPI_NONE, B)
        h1 = hs.c0.fwd(fused, ws.buf("fh.h1", (B, cl[0].weight.shape[0])))
        h2 = hs.c1.fwd(h1, ws.buf("fh.h2", (B, cl[4].weight.shape[0])))
        logits = hs.c2.fwd(h2, ws.buf("fh.logits", (B, cl[8].weight.shape[0])))
        return _finish(hs, ws, logits, labels, B, True)

    def _classify_backward(self, hs, dlogits, grads, ws):
        B = hs.B
        dh2 = ws.buf("fh.dh2", (B, hs.c2.K))
        hs.c2.bwd(ws, dlogits, dh2, grads)
        dh1 = ws.buf("fh.dh1", (B, hs.c1.K))
        hs.c1.bwd(ws, dh2, dh1, grads)
        dfused = ws.buf("fh.dfused", (B, hs.c0.K))
        hs.c0.bwd(ws, dh1, dfused, grads)
        return dfused

    def forward(self, text_features, audio_features, labels=None):
        dev = _require_gpu(self, "HybridFusionHead")
        t = text_features.to(device=dev, dtype=torch.float32).contiguous()
        a = audio_features.to(device=dev, dtype=torch.float32).contiguous()
        if labels is not None:
            labels = labels.to(dev)
            if labels.dim() == 1 and labels.dtype != torch.long:
                labels = (labels + 1).long()
            # nn.CrossEntropyLoss (early_fusion.py:346) refuses a target outside the classes; the loss kernel would
            # treat the row as one without a target.  (One read from the device: this head is a few launches on eight
            # rows, and the reference's loop reads the loss after every step anyway.)
            if labels.numel():
                lo, hi = int(labels.min()), int(labels.max())
                if lo < 0 or hi >= self.classifier[-1].weight.shape[0]:
                    raise IndexError(f"Target {lo if lo < 0 else hi} is out of bounds.")
        fl = self.fusion_layer
        if _needs_grad(self) or (torch.is_grad_enabled() and (t.requires_grad or a.requires_grad)):
            params = list(fl.parameters()) + list(self.classifier.parameters())
            _, wts, logits, probs, loss = _FusionFn.apply(fl, self, t, a, labels, *params)
            return {"loss": loss, "logits": logits, "probabilities": probs, "attention_weights": wts}
        ws = self.__dict__.setdefault("_ws", Workspace(dev))
        st = fusion_forward(fl, t, a, self.training, fl._seed(), ws)
        hs = self._classify(st.fused, labels, ws)
        loss = hs.loss.view(()).clone() if hs.loss is not None else None
        return {"loss": loss, "logits": hs.logits.clone(), "probabilities": hs.probs.clone(),
                "attention_weights": st.wts.view(-1, 2).clone()}

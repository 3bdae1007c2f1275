"""Fusion head and the three classifier stacks: launch sequences over the dense kernels.

'both'  : `CNN implementation/multimodal_model.py:216-225` (normaliser), `:263-285` (fusion), `:393-404`
'audio' : `:365-377` (Linear, BatchNorm1d, ReLU, Dropout x2, Linear)
'text'  : `:379-391` (Dropout, Linear, ReLU, LayerNorm, Dropout x2, Linear)
then softmax + mean cross-entropy (`:448-454`).  Dropout masks come from a counter-based generator inside
the kernels; the two modality-dropout decisions are host draws passed in as 0/1 device scalars.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from .ops import EPI_LN_RELU, EPI_NONE, EPI_RELU_LN
from .workspace import Grads, Workspace


class _Dense:
    """Bookkeeping for one Linear (+epilogue) so backward can mirror forward."""

    def __init__(self, ws, name, lin, norm, epi, B, drop_p=0.0, seed=0, step=None):
        self.lin, self.norm, self.epi, self.drop_p, self.seed, self.name = lin, norm, epi, drop_p, seed, name
        self.step = step if drop_p > 0 else None      # device-resident step counter mixed into the dropout seed
        self.N, self.K = lin.weight.shape
        ln = epi in (EPI_LN_RELU, EPI_RELU_LN, ops.EPI_LN, ops.EPI_LN_GELU)
        self.pre = ws.buf(name + ".pre", (B, self.N)) if epi != EPI_NONE else None
        self.stats = ws.buf(name + ".stats", 2 * B) if ln else None

    def fwd(self, X, Y):
        self.X = X
        g = self.norm.weight if self.norm is not None else None
        b = self.norm.bias if self.norm is not None else None
        ops.dense_fwd(X, self.lin.weight, self.lin.bias, g, b, Y, self.pre, self.stats, self.epi, self.drop_p,
                      self.seed, self.step)
        return Y

    def bwd(self, ws, dY, dX, grads: Grads, accumulate_dx=False):
        B = dY.shape[0]
        dPre = ws.buf(self.name + ".dpre", (B, self.N))
        ln = self.norm is not None
        dLn = ws.buf(self.name + ".dln", (B, self.N)) if ln else None
        from .audio import fork
        # the data gradient continues the chain on this stream; the parameter gradients only feed the optimiser
        # and go to a helper stream (joined by the caller with fork(...).join() / at the end of the step)
        ops.dense_bwd(dY, self.X, self.pre, self.stats, self.lin.weight,
                      self.norm.weight if ln else None, self.norm.bias if ln else None, dPre, dLn, dX,
                      None, None, None, None, self.epi, self.drop_p, self.seed, accumulate_dx, self.step)
        with fork(ws, dY.device, 2):
            ops.dense_bwd_w(dPre, dLn, self.X, self.pre, self.stats, grads.of(self.lin.weight),
                            grads.of(self.lin.bias), grads.of(self.norm.weight) if ln else None,
                            grads.of(self.norm.bias) if ln else None, self.epi)


def _finish(st, ws, logits, labels, B, need_grad):
    C = logits.shape[1]
    st.logits = logits
    st.probs = ws.buf("head.probs", (B, C))
    st.loss = None
    st.dlogits = None
    if labels is not None:
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.to(torch.int64)
        st.loss = ws.buf("head.ce", 1)
        st.dlogits = ws.buf("head.dlogits", (B, C)) if need_grad else None
    ops.softmax_ce(logits, labels, st.probs, st.loss, st.dlogits)
    return st


def _use_layers() -> bool:
    """MMSA_HEAD=layers selects the layer-by-layer launch sequence (kept for cross-checking the fused kernels)."""
    import os
    return os.environ.get("MMSA_HEAD", "") == "layers"


def head_forward_both(model, a_feat, t_feat, labels, training, keep_a, keep_t, drop_p, seed, need_grad,
                      ws: Workspace, logits_out=None, grad_scale: float = 1.0, step_state=None):
    """Fused path: the whole head is one launch (one workgroup per utterance).  When a gradient is wanted the launch
    also writes d(loss)/d(logits) (times ``grad_scale``), so the backward can follow at once; probabilities and the
    loss value are then computed beside it on a helper stream.  Otherwise softmax / CE follow on this stream.
    ``step_state`` (a one-element float tensor holding the optimiser's step count) is mixed into the dropout seed on
    the device, which is what lets a replayed launch sequence draw new masks every step."""
    if _use_layers():
        return head_forward_both_layers(model, a_feat, t_feat, labels, training, keep_a, keep_t, drop_p, seed,
                                        need_grad, ws, logits_out, step_state)
    B, A = a_feat.shape
    T = t_feat.shape[1]
    fn, fu, cl = model.feature_normalizer, model.fusion_layer, model.classifier
    Hd = fu.audio_projector[0].weight.shape[0]
    N0, N1, C = cl[0].weight.shape[0], cl[4].weight.shape[0], cl[8].weight.shape[0]
    p = drop_p if training else 0.0
    st = SimpleNamespace(kind="both", one_launch=True, B=B, dims=(B, A, T, Hd, N0, N1, C), p=p, seed=seed)
    logits = logits_out if logits_out is not None else ws.buf("head.logits", (B, C))
    t = {
        "a_feat": a_feat.contiguous(), "t_feat": t_feat.contiguous(),
        "an_g": fn.audio_norm.weight, "an_b": fn.audio_norm.bias, "a_scale": fn.audio_scale,
        "tn_g": fn.text_norm.weight, "tn_b": fn.text_norm.bias, "t_scale": fn.text_scale,
        "keep_a": keep_a, "keep_t": keep_t,
        "wpa": fu.audio_projector[0].weight, "bpa": fu.audio_projector[0].bias,
        "gpa": fu.audio_projector[1].weight, "bepa": fu.audio_projector[1].bias,
        "wpt": fu.text_projector[0].weight, "bpt": fu.text_projector[0].bias,
        "gpt": fu.text_projector[1].weight, "bept": fu.text_projector[1].bias,
        "watt": fu.attention[0].weight, "batt": fu.attention[0].bias,
        "w0": cl[0].weight, "b0": cl[0].bias, "g0": cl[1].weight, "be0": cl[1].bias,
        "w1": cl[4].weight, "b1": cl[4].bias, "g1": cl[5].weight, "be1": cl[5].bias,
        "w2": cl[8].weight, "b2": cl[8].bias,
        "an": ws.buf("head.an", (B, A)), "tn": ws.buf("head.tn", (B, T)),
        "an_stats": ws.buf("head.an_stats", 2 * B), "tn_stats": ws.buf("head.tn_stats", 2 * B),
        "pa_pre": ws.buf("head.pa.pre", (B, Hd)), "pa_stats": ws.buf("head.pa.stats", 2 * B),
        "pt_pre": ws.buf("head.pt.pre", (B, Hd)), "pt_stats": ws.buf("head.pt.stats", 2 * B),
        "P": ws.buf("head.P", (B, 2 * Hd)), "att": ws.buf("head.att_out", (B, 2)), "wts": ws.buf("head.wts", 2 * B),
        "fused": ws.buf("head.fused", (B, Hd)),
        "c0_pre": ws.buf("head.c0.pre", (B, N0)), "c0_stats": ws.buf("head.c0.stats", 2 * B),
        "h1": ws.buf("head.h1", (B, N0)),
        "c1_pre": ws.buf("head.c1.pre", (B, N1)), "c1_stats": ws.buf("head.c1.stats", 2 * B),
        "h2": ws.buf("head.h2", (B, N1)), "logits": logits,
    }
    if step_state is not None and p > 0:
        t["step"] = step_state
    st.tensors = t
    direct = labels is not None and need_grad
    if direct:
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            labels = labels.to(torch.int64)
        t["labels"] = labels
        t["dlogits"] = ws.buf("head.dlogits", (B, C))
    ops.head_both("fwd", ops.head_both_table(t), st.dims, p, seed, grad_scale)
    if not direct:
        return _finish(st, ws, logits, labels, B, need_grad)
    from .audio import fork
    st.logits, st.dlogits, st.grad_scaled = logits, t["dlogits"], True
    st.probs = ws.buf("head.probs", (B, C))
    st.loss = ws.buf("head.ce", 1)

    def loss_value():
        ops.softmax_ce(logits, labels, st.probs, st.loss, None)
    if ws.persistent:
        # launched beside the backward (helper stream 2, after the critical-path kernels have been enqueued: a
        # captured launch sequence keeps a node's first-captured successor on that node's queue)
        st.deferred = [loss_value]
    else:
        loss_value()
    return st


def head_backward_both(model, st, dlogits, grads: Grads, ws: Workspace):
    """Returns (d_audio_features, d_text_features)."""
    if not getattr(st, "one_launch", False):
        return head_backward_both_layers(model, st, dlogits, grads, ws)
    from .audio import fork
    B, A, T, Hd, N0, N1, C = st.dims
    fn, fu, cl = model.feature_normalizer, model.fusion_layer, model.classifier
    t = st.tensors
    t.update({
        "dlogits": dlogits.contiguous(),
        "c1_dpre": ws.buf("head.c1.dpre", (B, N1)), "c1_dln": ws.buf("head.c1.dln", (B, N1)),
        "c0_dpre": ws.buf("head.c0.dpre", (B, N0)), "c0_dln": ws.buf("head.c0.dln", (B, N0)),
        "att_dpre": ws.buf("head.att.dpre", (B, 2)),
        "pa_dpre": ws.buf("head.pa.dpre", (B, Hd)), "pa_dln": ws.buf("head.pa.dln", (B, Hd)),
        "pt_dpre": ws.buf("head.pt.dpre", (B, Hd)), "pt_dln": ws.buf("head.pt.dln", (B, Hd)),
        "an_dln": ws.buf("head.an_dln", (B, A)), "tn_dln": ws.buf("head.tn_dln", (B, T)),
        "a_dspart": ws.buf("head.a_dspart", B), "t_dspart": ws.buf("head.t_dspart", B),
        "d_afeat": ws.buf("head.d_afeat", (B, A)), "d_tfeat": ws.buf("head.d_tfeat", (B, T)),
    })
    g = grads.of
    t.update({
        "dwpa": g(fu.audio_projector[0].weight), "dbpa": g(fu.audio_projector[0].bias),
        "dgpa": g(fu.audio_projector[1].weight), "dbepa": g(fu.audio_projector[1].bias),
        "dwpt": g(fu.text_projector[0].weight), "dbpt": g(fu.text_projector[0].bias),
        "dgpt": g(fu.text_projector[1].weight), "dbept": g(fu.text_projector[1].bias),
        "dwatt": g(fu.attention[0].weight), "dbatt": g(fu.attention[0].bias),
        "dw0": g(cl[0].weight), "db0": g(cl[0].bias), "dg0": g(cl[1].weight), "dbe0": g(cl[1].bias),
        "dw1": g(cl[4].weight), "db1": g(cl[4].bias), "dg1": g(cl[5].weight), "dbe1": g(cl[5].bias),
        "dw2": g(cl[8].weight), "db2": g(cl[8].bias),
        "dan_g": g(fn.audio_norm.weight), "dan_b": g(fn.audio_norm.bias), "da_scale": g(fn.audio_scale),
        "dtn_g": g(fn.text_norm.weight), "dtn_b": g(fn.text_norm.bias), "dt_scale": g(fn.text_scale),
    })
    table = ops.head_both_table(t)
    if not ws.persistent:
        # autograd path: the gradient tensors are needed as addresses only, and the table has them.  Letting go of them
        # here leaves autograd the sole owner of what it is handed, so it adopts the tensors instead of copying them
        for k in ("dwpa", "dbpa", "dgpa", "dbepa", "dwpt", "dbpt", "dgpt", "dbept", "dwatt", "dbatt", "dw0", "db0", "dg0",
                  "dbe0", "dw1", "db1", "dg1", "dbe1", "dw2", "db2", "dan_g", "dan_b", "da_scale", "dtn_g", "dtn_b",
                  "dt_scale"):
            del t[k]
    ops.head_both("bwd", table, st.dims, st.p, st.seed)
    # the parameter gradients only feed the optimiser: helper stream 2 (joined at the end of the step), enqueued by
    # the caller through st.side_work once the data-gradient chain has been launched
    point = fork(ws, dlogits.device, 2).mark()

    def param_grads():
        with point:
            ops.head_both("bwd_params", table, st.dims)
    st.side_work = [param_grads]
    return t["d_afeat"], t["d_tfeat"]


def head_forward_both_layers(model, a_feat, t_feat, labels, training, keep_a, keep_t, drop_p, seed, need_grad,
                             ws: Workspace, logits_out=None, step_state=None):
    B = a_feat.shape[0]
    fn, fu, cl = model.feature_normalizer, model.fusion_layer, model.classifier
    st = SimpleNamespace(kind="both", B=B, a_feat=a_feat, t_feat=t_feat, keep_a=keep_a, keep_t=keep_t)
    p = drop_p if training else 0.0
    st.an = ws.buf("head.an", a_feat.shape)
    st.tn = ws.buf("head.tn", t_feat.shape)
    st.an_stats = ws.buf("head.an_stats", 2 * B)
    st.tn_stats = ws.buf("head.tn_stats", 2 * B)
    ops.ln_scale_fwd(a_feat, fn.audio_norm.weight, fn.audio_norm.bias, fn.audio_scale, keep_a, st.an, st.an_stats)
    ops.ln_scale_fwd(t_feat, fn.text_norm.weight, fn.text_norm.bias, fn.text_scale, keep_t, st.tn, st.tn_stats)
    Hd = fu.audio_projector[0].weight.shape[0]
    st.Hd = Hd
    st.P = ws.buf("head.P", (B, 2 * Hd))
    st.pa = _Dense(ws, "head.pa", fu.audio_projector[0], fu.audio_projector[1], EPI_LN_RELU, B, p, seed + 1, step_state)
    st.pt = _Dense(ws, "head.pt", fu.text_projector[0], fu.text_projector[1], EPI_LN_RELU, B, p, seed + 2, step_state)
    st.pa.fwd(st.an, st.P[:, :Hd])
    st.pt.fwd(st.tn, st.P[:, Hd:])
    st.att = _Dense(ws, "head.att", fu.attention[0], None, EPI_NONE, B)
    st.att_out = st.att.fwd(st.P, ws.buf("head.att_out", (B, 2)))
    st.wts = ws.buf("head.wts", 2 * B)
    st.fused = ws.buf("head.fused", (B, Hd))
    ops.fusion_mix_fwd(st.P, st.att_out, st.wts, st.fused)
    st.c0 = _Dense(ws, "head.c0", cl[0], cl[1], EPI_LN_RELU, B, p, seed + 3, step_state)
    st.c1 = _Dense(ws, "head.c1", cl[4], cl[5], EPI_LN_RELU, B, p / 2, seed + 4, step_state)
    st.c2 = _Dense(ws, "head.c2", cl[8], None, EPI_NONE, B)
    h1 = st.c0.fwd(st.fused, ws.buf("head.h1", (B, cl[0].weight.shape[0])))
    h2 = st.c1.fwd(h1, ws.buf("head.h2", (B, cl[4].weight.shape[0])))
    logits = st.c2.fwd(h2, logits_out if logits_out is not None else ws.buf("head.logits", (B, cl[8].weight.shape[0])))
    return _finish(st, ws, logits, labels, B, need_grad)


def head_backward_both_layers(model, st, dlogits, grads: Grads, ws: Workspace):
    """Returns (d_audio_features, d_text_features)."""
    B, Hd = st.B, st.Hd
    fn = model.feature_normalizer
    dh2 = ws.buf("head.dh2", (B, st.c2.K))
    st.c2.bwd(ws, dlogits, dh2, grads)
    dh1 = ws.buf("head.dh1", (B, st.c1.K))
    st.c1.bwd(ws, dh2, dh1, grads)
    dfused = ws.buf("head.dfused", (B, Hd))
    st.c0.bwd(ws, dh1, dfused, grads)
    dP = ws.buf("head.dP", (B, 2 * Hd))
    datt = ws.buf("head.datt", (B, 2))
    ops.fusion_mix_bwd(dfused, st.P, st.wts, dP, datt)
    st.att.bwd(ws, datt, dP, grads, accumulate_dx=True)
    d_an = ws.buf("head.d_an", st.an.shape)
    d_tn = ws.buf("head.d_tn", st.tn.shape)
    st.pa.bwd(ws, dP[:, :Hd], d_an, grads)
    st.pt.bwd(ws, dP[:, Hd:], d_tn, grads)
    d_a = ws.buf("head.d_afeat", st.a_feat.shape)
    d_t = ws.buf("head.d_tfeat", st.t_feat.shape)
    sa = ws.buf("head.ln_a_scratch", B * st.a_feat.shape[1] + B)
    stx = ws.buf("head.ln_t_scratch", B * st.t_feat.shape[1] + B)
    ops.ln_scale_bwd(d_an, st.a_feat, fn.audio_norm.weight, fn.audio_norm.bias, fn.audio_scale, st.keep_a,
                     st.an_stats, d_a, grads.of(fn.audio_norm.weight), grads.of(fn.audio_norm.bias),
                     grads.of(fn.audio_scale), sa)
    ops.ln_scale_bwd(d_tn, st.t_feat, fn.text_norm.weight, fn.text_norm.bias, fn.text_scale, st.keep_t,
                     st.tn_stats, d_t, grads.of(fn.text_norm.weight), grads.of(fn.text_norm.bias),
                     grads.of(fn.text_scale), stx)
    return d_a, d_t


def head_forward_audio(model, feat, labels, training, drop_p, seed, need_grad, ws: Workspace, logits_out=None,
                       step_state=None):
    B = feat.shape[0]
    cl = model.classifier
    p = drop_p if training else 0.0
    step = step_state if p > 0 else None
    if B == 1 and (cl[2].training or cl[6].training):
        # what nn.BatchNorm1d says to a single row in training mode (multimodal_model.py:369-375): its variance is zero
        raise ValueError(f"Expected more than 1 value per channel when training, got input size "
                         f"torch.Size([1, {cl[1].weight.shape[0]}])")
    # (the two BatchNorm1d layers go by their own flags, as in torch; dropout by the model's)
    st = SimpleNamespace(kind="audio", B=B, feat=feat, training=training, p=p, seed=seed, step=step,
                         bn_training=(cl[2].training, cl[6].training))
    st.l1 = _Dense(ws, "head.a1", cl[1], None, EPI_NONE, B)
    st.l2 = _Dense(ws, "head.a5", cl[5], None, EPI_NONE, B)
    st.l3 = _Dense(ws, "head.a9", cl[9], None, EPI_NONE, B)
    st.z1 = st.l1.fwd(feat, ws.buf("head.z1", (B, cl[1].weight.shape[0])))
    st.y1 = ws.buf("head.y1", st.z1.shape)
    st.cf1 = ws.buf("head.cf1", 4 * st.z1.shape[1])
    bn = cl[2]
    ops.bn1d_relu_fwd(st.z1, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                      st.bn_training[0], st.y1, st.cf1, p, seed + 1, step)
    st.z2 = st.l2.fwd(st.y1, ws.buf("head.z2", (B, cl[5].weight.shape[0])))
    st.y2 = ws.buf("head.y2", st.z2.shape)
    st.cf2 = ws.buf("head.cf2", 4 * st.z2.shape[1])
    bn = cl[6]
    ops.bn1d_relu_fwd(st.z2, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                      st.bn_training[1], st.y2, st.cf2, p / 2, seed + 2, step)
    logits = st.l3.fwd(st.y2, logits_out if logits_out is not None else ws.buf("head.logits", (B, cl[9].weight.shape[0])))
    return _finish(st, ws, logits, labels, B, need_grad)


def head_backward_audio(model, st, dlogits, grads: Grads, ws: Workspace):
    B = st.B
    cl = model.classifier
    dy2 = ws.buf("head.dy2", st.y2.shape)
    st.l3.bwd(ws, dlogits, dy2, grads)
    dz2 = ws.buf("head.dz2", st.z2.shape)
    ops.bn1d_relu_bwd(dy2, st.z2, st.cf2, dz2, grads.of(cl[6].weight), grads.of(cl[6].bias), st.bn_training[1],
                      st.p / 2, st.seed + 2, st.step)
    dy1 = ws.buf("head.dy1", st.y1.shape)
    st.l2.bwd(ws, dz2, dy1, grads)
    dz1 = ws.buf("head.dz1", st.z1.shape)
    ops.bn1d_relu_bwd(dy1, st.z1, st.cf1, dz1, grads.of(cl[2].weight), grads.of(cl[2].bias), st.bn_training[0], st.p,
                      st.seed + 1, st.step)
    dfeat = ws.buf("head.dfeat", st.feat.shape)
    st.l1.bwd(ws, dz1, dfeat, grads)
    return dfeat


def head_forward_text(model, feat, labels, training, drop_p, seed, need_grad, ws: Workspace, logits_out=None,
                      step_state=None):
    B = feat.shape[0]
    cl = model.classifier
    p = drop_p if training else 0.0
    step = step_state if p > 0 else None
    st = SimpleNamespace(kind="text", B=B, feat=feat, p=p, seed=seed, step=step)
    x0 = feat
    if p > 0.0:
        x0 = ws.buf("head.x0", feat.shape)
        ops.dropout(feat, x0, p, seed + 9, step)
    st.l1 = _Dense(ws, "head.t1", cl[1], cl[3], EPI_RELU_LN, B, p, seed + 1, step)
    st.l2 = _Dense(ws, "head.t5", cl[5], cl[7], EPI_RELU_LN, B, p, seed + 2, step)
    st.l3 = _Dense(ws, "head.t9", cl[9], None, EPI_NONE, B)
    h1 = st.l1.fwd(x0, ws.buf("head.h1", (B, cl[1].weight.shape[0])))
    h2 = st.l2.fwd(h1, ws.buf("head.h2", (B, cl[5].weight.shape[0])))
    logits = st.l3.fwd(h2, logits_out if logits_out is not None else ws.buf("head.logits", (B, cl[9].weight.shape[0])))
    return _finish(st, ws, logits, labels, B, need_grad)


def head_backward_text(model, st, dlogits, grads: Grads, ws: Workspace):
    B = st.B
    dh2 = ws.buf("head.dh2", (B, st.l3.K))
    st.l3.bwd(ws, dlogits, dh2, grads)
    dh1 = ws.buf("head.dh1", (B, st.l2.K))
    st.l2.bwd(ws, dh2, dh1, grads)
    dx0 = ws.buf("head.dx0", st.feat.shape)
    st.l1.bwd(ws, dh1, dx0, grads)
    if st.p > 0.0:
        dfeat = ws.buf("head.dfeat", st.feat.shape)
        ops.dropout(dx0, dfeat, st.p, st.seed + 9, st.step)
        return dfeat
    return dx0

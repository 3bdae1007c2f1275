"""Text branch: launch sequences (gather -> per-width conv -> bias+ReLU+max over time) and the sparse backward.

Computes what `CNN implementation/multimodal_model.py:173-199` (TextModel.forward) computes; the attention
mask takes no part in the arithmetic there, so it is accepted and ignored here too.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from .workspace import Grads, Workspace


def text_forward(m, input_ids: torch.Tensor, need_grad: bool, ws: Workspace, err=None):
    if input_ids.dim() != 2:
        raise ValueError("input_ids must be [B, L]")
    ids = input_ids.contiguous()
    if ids.dtype != torch.int64:
        ids = ids.to(torch.int64)
    B, L = ids.shape
    table = m.embedding.weight
    V, E = table.shape
    ksizes = [c.weight.shape[2] for c in m.convs]
    F = m.convs[0].weight.shape[0]
    if E % 32 or F % 64 or any(c.weight.shape[0] != F for c in m.convs):
        raise NotImplementedError("HIP text branch needs embedding_dim % 32 == 0 and num_filters % 64 == 0")
    if max(ksizes) > 7 or min(ksizes) < 2 or L < max(ksizes):
        raise NotImplementedError("HIP text branch supports filter sizes 2..7 and sequences at least that long")
    if need_grad and (len(ksizes) > 4 or E % 64):
        raise NotImplementedError("HIP text branch, backward: at most four filter widths and embedding_dim % 64 == 0 "
                                  "(the forward alone, under torch.no_grad(), takes more widths and embedding_dim % 32)")
    st = SimpleNamespace(B=B, L=L, E=E, F=F, ksizes=ksizes, ids=ids)
    st.emb = ws.buf("text.emb", (B, L, E))
    # sticky flag: set by the gather / the backward scatter when a token id has no row in the table (a persistent
    # workspace creates it zeroed once; check_ids reads and clears it)
    st.err = err if err is not None else ws.buf("text.err", 1, torch.int32, zero=not ws.persistent)
    ops.embedding_gather(ids, table, st.emb, st.err)
    probe = ws.objs.get("text.err_probe")
    if probe is not None:
        # the flag as the gather left it goes to pinned host memory at once, with an event behind it: the caller can
        # look at it when it has enqueued the rest of the forward pass, without waiting for that pass to finish
        probe[0].copy_(st.err, non_blocking=True)
        probe[1].record()
    st.pad = m.embedding.padding_idx if m.embedding.padding_idx is not None else -1
    n = len(ksizes)
    st.feat = ws.buf("text.feat", (B, n * F))
    st.arg = ws.buf("text.arg", (B, n * F), torch.int32)
    st.wps = [ws.buf(f"text.wp{i}", conv.weight.numel()) for i, conv in enumerate(m.convs)]
    plan = ws.objs.get("text.packplan")
    if plan is None:
        plan = ops.PackPlan()
        if ws.persistent:
            ws.objs["text.packplan"] = plan
    plan.run([(conv.weight, wp, None) for conv, wp in zip(m.convs, st.wps)])
    ys = [ws.buf(f"text.y{i}", (B, L, F)) for i in range(n)]
    taps = [(0, j) for j in range(max(ksizes))]
    if n <= 4:
        # the filter widths share the input and differ only in tap count: one launch for all of them
        ops.conv_fwd_multi(st.emb, st.wps, ys, ksizes, B, 1, L, E, F, taps)
        ops.bias_relu_maxtime_multi(ys, [c.bias for c in m.convs], [L - k + 1 for k in ksizes], st.feat, st.arg,
                                    B, L, F)
    else:
        for i, (conv, k) in enumerate(zip(m.convs, ksizes)):
            ops.conv_fwd(st.emb, st.wps[i], ys[i], None, B, 1, L, E, F, taps[:k])
            ops.bias_relu_maxtime(ys[i], conv.bias, st.feat, st.arg, B, L, L - k + 1, F, n * F, i * F)
    if need_grad:
        # what the backward needs to know about repeated tokens depends on the ids only: sorted here, beside the
        # forward pass, where this branch has time to spare (id table: zero once, every launch leaves it zero again)
        st.groups = ws.buf("text.groups", 5 * B * L, torch.int32)
        ops.token_groups(ids, V, st.pad, st.groups, ws.buf("text.id_table", 2 * V, torch.int32, zero=not ws.persistent))
        host = ws.objs.get("text.err_host")
        if host is not None:
            # the sticky flag goes to pinned host memory in every step (a 4-byte copy on this branch's stream, part of
            # a captured step): the host can look at it without synchronising
            host.copy_(st.err, non_blocking=True)
    return st.feat, st


def _raise_for(flag: int):
    """The sticky flag word: bit 0 a token id outside the embedding table (``nn.Embedding`` raises this IndexError in
    the reference, multimodal_model.py:179), bit 1 a class label outside [0, C) (``nn.CrossEntropyLoss``,
    multimodal_model.py:452-454)."""
    if flag & 1:
        raise IndexError("index out of range in self")
    if flag & 2:
        raise IndexError("Target is out of bounds.")


def check_ids(err: torch.Tensor):
    """Raises what the reference raises when a token id was outside the table or a label outside the classes.  Reads
    one int from the device (synchronises) and clears it."""
    flag = int(err.item())
    if flag:
        err.zero_()
        _raise_for(flag)


def check_ids_probe(probe):
    """The same from the copy that the forward pass left in pinned host memory behind the gather (``probe``: the pinned
    int and the event recorded behind its copy).  Waits for the head of the forward pass only."""
    probe[1].synchronize()
    flag = int(probe[0][0])
    if flag:
        probe[0].zero_()
        _raise_for(flag)


def check_labels_host(labels: torch.Tensor, num_classes: int):
    """The label check for labels that are still in host memory (no device round trip)."""
    if labels.numel():
        lo, hi = int(labels.min()), int(labels.max())
        if lo < 0 or hi >= num_classes:
            raise IndexError(f"Target {lo if lo < 0 else hi} is out of bounds.")


def check_ids_host(input_ids: torch.Tensor, vocab: int):
    """The same check for token ids that are still in host memory (no device round trip)."""
    if input_ids.numel() and (int(input_ids.min()) < 0 or int(input_ids.max()) >= vocab):
        raise IndexError("index out of range in self")


def text_backward(m, st, dfeat: torch.Tensor, grads: Grads, ws: Workspace):
    table = m.embedding.weight
    dtable = grads.of(table)
    if not getattr(st, "dtable_zeroed", False):
        dtable.zero_()    # the kernels write the rows of this batch's tokens only
    ops.text_bwd(dfeat.contiguous(), st.feat, st.arg, st.emb, st.ids, st.ksizes, st.wps,
                 [grads.of(c.weight) for c in m.convs], [grads.of(c.bias) for c in m.convs], dtable,
                 st.B, st.L, st.E, st.F, padding_idx=st.pad, err=st.err, drow=ws.buf("text.drow", (st.B * st.L, st.E)),
                 groups=st.groups)

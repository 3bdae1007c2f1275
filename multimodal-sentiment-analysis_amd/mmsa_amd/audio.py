"""Audio branch: launch sequences for forward and backward over the HIP kernels.

Computes what `CNN implementation/multimodal_model.py:118-129` (AudioModel.forward) and `:64-70`
(ResidualBlock.forward) compute, on channels-last activations.  Nothing here does arithmetic: it sizes
buffers and orders kernel launches.  BatchNorm statistics ride on the convolution epilogues, the
conv->BN->ReLU->pool stem never materialises its full-resolution output, and SE / residual add / ReLU /
max-pool are one kernel.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from .workspace import Grads, Workspace


_SIDE_STREAMS = {}
import os as _os
FORK_WGRAD = _os.environ.get("MMSA_FORK_WGRAD", "1") != "0"
FORK_ANY = _os.environ.get("MMSA_FORK", "1") != "0"     # 0: the whole step on one stream (diagnostics)
# 0: BatchNorm-backward reductions of a block's first BatchNorm by a pass of their own instead of in the epilogue of
# the data-gradient convolution that produces their input (A/B aid)
FUSE_BN_BWD_STATS = _os.environ.get("MMSA_FUSE_BN_BWD_STATS", "1") != "0"
# 0: the global average pool behind the last block (and its backward) as launches of their own instead of inside the
# block's pooling kernel / its backward (A/B aid)
FUSE_GAP = _os.environ.get("MMSA_FUSE_GAP", "1") != "0"
# 1: the squeeze-excite gate inside the pooling kernel that consumes it (0: a launch of its own; A/B aid)
FUSE_SE_GATE = _os.environ.get("MMSA_FUSE_SE_GATE", "1") != "0"
# 1: a block's first 3x3 convolution and its 1x1 shortcut share one launch (and one BatchNorm coefficient launch)
FWD_PAIR = _os.environ.get("MMSA_FWD_PAIR", "1") != "0"
# 1: inference folds bn1 + ReLU into the epilogue of a block's first convolution (0: separate pass, as in training)
FOLD_EVAL_RELU = _os.environ.get("MMSA_FOLD_EVAL_RELU", "1") != "0"
# Weight-gradient kernels: "slabs" (default) = the DMA-fed second generation (one 512-thread workgroup per CU writes a
# partial slab; a block's slabs are summed by one launch behind the block's last weight-gradient kernel, or with
# MMSA_WGRAD_REDUCE=end by one launch for all layers at the end of the backward pass); "old" = the first generation
# (two 256-thread workgroups per CU, one reduction launch per layer), which also serves maps too wide for a strip in
# LDS.  Alone, the second generation is faster on every layer (50 vs 71 us with its reduction on the 256 -> 256 layer at
# batch 32) and writes fewer partial sums; inside the step the two are level (batch 32, one GPU: 1.413 vs 1.405 ms):
# what the helper stream's kernels hide behind is the data-gradient chain's small element-wise kernels, and the
# slower generation hides as much more as it is slower (measured without any weight gradient: 1.211 ms; with all of
# them behind the last data-gradient convolution: 1.585 / 1.511 ms; a single reduction at the end: 1.437 ms).
WGRAD = _os.environ.get("MMSA_WGRAD", "slabs")
WGRAD_REDUCE_PER_BLOCK = _os.environ.get("MMSA_WGRAD_REDUCE", "block") == "block"
# 1: every weight-gradient launch waits until the last data-gradient convolution has been launched, then the whole
# chain runs beside the stem's backward.  0 (default): each launch from its own fork point, as early as its inputs
# exist.
WGRAD_DEFER = _os.environ.get("MMSA_WGRAD_DEFER", "0") != "0"


def side_stream(device, which: int = 0) -> torch.cuda.Stream:
    """Per-device helper streams.  Work that is off the critical path (weight gradients; the text branch) is
    enqueued there between explicit fork / join points, so under-filled launches overlap instead of queueing."""
    key = (torch.device(device).index, which)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _SIDE_STREAMS[key] = s
    return s


class fork:
    """``with fork(ws, device) as f: ...`` runs the body on a helper stream that first waits for the current
    stream (fork point); ``f.join()`` later makes the current stream wait for it.  Only persistent workspaces
    fork: their buffers are never freed, so no allocator bookkeeping across streams is needed; with throw-away
    buffers (autograd path) everything stays on one stream."""

    def __init__(self, ws, device, which: int = 0, enable: bool = True):
        self.side = side_stream(device, which) if ws.persistent and enable and FORK_ANY else None
        self.ctx = None
        self.point = None

    def mark(self):
        """Fixes the fork point at the current position of the current stream; work enqueued there afterwards is
        not waited for by the helper stream.  Lets the caller launch the critical-path kernel first: a captured
        launch sequence tends to keep a node's first-captured successor on that node's queue."""
        if self.side is not None:
            self.point = torch.cuda.Event()
            self.point.record(torch.cuda.current_stream())
        return self

    def __enter__(self):
        if self.side is not None:
            if self.point is not None:
                self.side.wait_event(self.point)
            else:
                self.side.wait_stream(torch.cuda.current_stream())
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False

    def join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)


def _bn_args(bn):
    return bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked


def _one_mode(flags, what: str) -> bool:
    flags = set(flags)
    if len(flags) > 1:
        raise NotImplementedError(f"{what}: some BatchNorm layers are in training mode and some in eval mode; the HIP "
                                  "kernels run all of them one way")
    return flags.pop()


def block_bn_training(blk) -> bool:
    """Whether a ResidualBlock's BatchNorm layers use batch statistics.  It is their own ``training`` flags that
    decide, as in torch: the usual way to freeze BatchNorm (``.eval()`` on the BatchNorm modules of a model that is
    otherwise in training mode) leaves the enclosing modules' flags alone."""
    flags = [blk.bn1.training, blk.bn2.training]
    if len(blk.shortcut) > 0:
        flags.append(blk.shortcut[1].training)
    return _one_mode(flags, "ResidualBlock")


def bn_training(m) -> bool:
    """The same for the whole audio branch (stem + three blocks: ten layers)."""
    return _one_mode([m.conv1[1].training] + [block_bn_training(b) for b in (m.res_block1, m.res_block2, m.res_block3)],
                     "AudioModel")


def _pack_bufs(ws: Workspace, name: str, w: torch.Tensor, need_dgrad: bool):
    f = ws.buf(name + ".wf", w.numel())
    d = ws.buf(name + ".wd", w.numel()) if need_dgrad else None
    return f, d


def pack_all(m, need_grad: bool, ws: Workspace):
    """Forward / data-gradient layouts of all seven convolution weights, re-packed in ONE launch."""
    out, triples = {}, []
    for i, blk in enumerate((m.res_block1, m.res_block2, m.res_block3)):
        for tag, conv in (("c1", blk.conv1), ("c2", blk.conv2), ("sc", blk.shortcut[0])):
            name = f"rb{i + 1}.{tag}"
            f, d = _pack_bufs(ws, name, conv.weight, need_grad)
            out[name] = (f, d)
            triples.append((conv.weight, f, d))
    plan = ws.objs.get("audio.packplan")
    if plan is None:
        plan = ops.PackPlan()
        if ws.persistent:
            ws.objs["audio.packplan"] = plan
    plan.run(triples)
    return out


def _conv_bn(ws, name, X, wf, bn, B, H, W, Cin, Cout, taps, training, want_bsum):
    """conv (+ statistics epilogue when needed) followed by the BatchNorm coefficient kernel."""
    Y = ws.buf(name + ".y", (B, H, W, Cout))
    coeff = ws.buf(name + ".coeff", 4 * Cout)
    need_stats = training or want_bsum
    rows, tiles = ops.conv_stats_rows(B, H, W, Cout)
    stats = ws.buf(name + ".stats", rows * 2 * Cout) if need_stats else None
    ops.conv_fwd(X, wf, Y, stats, B, H, W, Cin, Cout, taps)
    bsum = ws.buf(name + ".bsum", (B, Cout)) if want_bsum else None
    g, b, rm, rv, nbt = _bn_args(bn)
    ops.bn_finalize(stats, rows, tiles, B * H * W, g, b, rm, rv, nbt, training, coeff, bsum, Cout)
    return Y, coeff, bsum


def _conv_pair_bn(ws, name, X, w1f, wsf, bn1, bns, B, H, W, Cin, Cout, training, fold_relu=False):
    """conv1 (3x3) and the shortcut (1x1, the centre tap of the same offset table) of a residual block in one launch,
    then both BatchNorm coefficient sets in one launch.  fold_relu (inference only: the coefficients come from the
    running statistics, so they exist before the convolution): the first result is stored as relu(bn1(conv1)) by the
    convolution's own epilogue and returned in place of the raw convolution."""
    Ys = ws.buf(name + ".sc.y", (B, H, W, Cout))
    c1 = ws.buf(name + ".c1.coeff", 4 * Cout)
    cs = ws.buf(name + ".sc.coeff", 4 * Cout)
    rows, tiles = ops.conv_stats_rows(B, H, W, Cout)
    if fold_relu:
        assert not training
        H1 = ws.buf(name + ".h1", (B, H, W, Cout))
        ops.bn_finalize_pair([None, None], rows, tiles, B * H * W, [_bn_args(bn1), _bn_args(bns)], False, [c1, cs],
                             Cout)
        ops.conv_fwd_multi(X, [w1f, wsf], [H1, Ys], [9, 1], B, H, W, Cin, Cout, ops.TAPS3X3, tap0_list=[0, 4],
                           bnrelu_list=[c1, None])
        return H1, c1, Ys, cs
    Y1 = ws.buf(name + ".c1.y", (B, H, W, Cout))
    st = [ws.buf(name + ".c1.stats", rows * 2 * Cout), ws.buf(name + ".sc.stats", rows * 2 * Cout)] if training else None
    ops.conv_fwd_multi(X, [w1f, wsf], [Y1, Ys], [9, 1], B, H, W, Cin, Cout, ops.TAPS3X3, tap0_list=[0, 4],
                       stats_list=st)
    ops.bn_finalize_pair(st or [None, None], rows, tiles, B * H * W, [_bn_args(bn1), _bn_args(bns)], training,
                         [c1, cs], Cout)
    return Y1, c1, Ys, cs


def block_forward(blk, name, a_in, B, H, W, Cin, Cout, training, need_grad, ws: Workspace, packs, hook=None,
                  mean_out=None):
    """One ResidualBlock + the 2x2 max-pool that follows it.  a_in: [B,H,W,Cin] -> [B,H/2,W/2,Cout]."""
    s = SimpleNamespace(B=B, H=H, W=W, Cin=Cin, Cout=Cout, a_in=a_in, name=name)
    s.w1f, s.w1d = packs[name + ".c1"]
    s.w2f, s.w2d = packs[name + ".c2"]
    s.wsf, s.wsd = packs[name + ".sc"]
    fsc = None
    fold = FWD_PAIR and FOLD_EVAL_RELU and not training and not need_grad
    if fold:
        s.c1 = None
        s.h1, s.coeff1, s.sc, s.coeffs = _conv_pair_bn(ws, name, a_in, s.w1f, s.wsf, blk.bn1, blk.shortcut[1], B, H, W,
                                                       Cin, Cout, training, fold_relu=True)
    elif FWD_PAIR:
        s.c1, s.coeff1, s.sc, s.coeffs = _conv_pair_bn(ws, name, a_in, s.w1f, s.wsf, blk.bn1, blk.shortcut[1], B, H, W,
                                                       Cin, Cout, training)
    else:
        # the 1x1 shortcut only meets the main chain at the fused add: it runs beside conv1 -> bn1 -> conv2
        with fork(ws, a_in.device) as fsc:
            s.sc, s.coeffs, _ = _conv_bn(ws, name + ".sc", a_in, s.wsf, blk.shortcut[1], B, H, W, Cin, Cout,
                                         ops.TAPS1X1, training, False)
        s.c1, s.coeff1, _ = _conv_bn(ws, name + ".c1", a_in, s.w1f, blk.bn1, B, H, W, Cin, Cout, ops.TAPS3X3,
                                     training, False)
    if hook is not None:
        hook()          # the caller's side work, enqueued behind this block's first convolution
    if not fold:
        s.h1 = ws.buf(name + ".h1", (B, H, W, Cout))
        ops.bn_relu_apply(s.c1, s.coeff1, s.h1, B * H * W, Cout)
    s.c2, s.coeff2, s.bsum2 = _conv_bn(ws, name + ".c2", s.h1, s.w2f, blk.bn2, B, H, W, Cout, Cout, ops.TAPS3X3,
                                       training, True)
    W1, W2 = blk.se.fc[0].weight, blk.se.fc[2].weight
    Ch = W1.shape[0]
    s.gap = ws.buf(name + ".gap", (B, Cout))
    s.hid = ws.buf(name + ".hid", (B, Ch))
    s.gate = ws.buf(name + ".gate", (B, Cout))
    fused_gate = FUSE_SE_GATE and Cout <= 256 and Ch <= 16 and (mean_out is None or Cout % 64 == 0)
    if not fused_gate:
        ops.se_gate(s.bsum2, s.coeff2, W1, W2, H * W, s.gap, s.hid, s.gate, B, Cout)
    if fsc is not None:
        fsc.join()
    if mean_out is not None:
        # the branch's last block: only the mean of its pooled output is wanted (the global average pool follows)
        s.out = None
        if fused_gate:
            ops.res_se_gate_relu_pool(s.c2, s.sc, s.coeff2, s.coeffs, s.bsum2, W1, W2, s.gap, s.hid, s.gate, None,
                                      mean_out, B, H, W, Cout)
        else:
            ops.res_gate_relu_pool_mean(s.c2, s.sc, s.coeff2, s.coeffs, s.gate, mean_out, B, H, W, Cout)
        return s
    s.out = ws.buf(name + ".out", (B, H // 2, W // 2, Cout))
    if fused_gate:
        ops.res_se_gate_relu_pool(s.c2, s.sc, s.coeff2, s.coeffs, s.bsum2, W1, W2, s.gap, s.hid, s.gate, s.out, None,
                                  B, H, W, Cout)
    else:
        ops.res_gate_relu_pool(s.c2, s.sc, s.coeff2, s.coeffs, s.gate, s.out, B, H, W, Cout)
    return s


def block_backward(blk, s, d_out, training, grads: Grads, ws: Workspace, need_dx: bool, reduce_list=None,
                   deferred=None, d_out_is_mean=False):
    """d_out: [B,H/2,W/2,Cout] -> gradient w.r.t. the block input [B,H,W,Cin] (or None).

    The three weight gradients (and the squeeze-excite ones) feed nothing but the optimiser: they run on a helper
    stream from fork points fixed before the data-gradient kernels, which are launched first."""
    B, H, W, Cin, Cout, name = s.B, s.H, s.W, s.Cin, s.Cout, s.name
    W1, W2 = blk.se.fc[0].weight, blk.se.fc[2].weight
    d_c2 = ws.buf(name + ".d_c2", (B, H, W, Cout))
    d_sc = ws.buf(name + ".d_sc", (B, H, W, Cout))
    scratch = ws.buf(name + ".bwd_scratch", ops.resblock_bwd_scratch(B, H, W, Cout))
    bn2, bns = blk.bn2, blk.shortcut[1]
    ops.resblock_bwd_ew(d_out, s.c2, s.sc, s.coeff2, s.coeffs, s.gate, s.hid, s.gap, s.bsum2, W1, W2, d_c2, d_sc,
                        grads.of(bn2.weight), grads.of(bn2.bias), grads.of(bns.weight), grads.of(bns.bias),
                        None, None, scratch, B, H, W, Cout, training, dout_is_mean=d_out_is_mean)
    d_h1 = ws.buf(name + ".d_h1", (B, H, W, Cout))
    d_c1 = d_h1                                   # bn_relu_bwd works in place
    d_in = ws.buf(name + ".d_in", (B, H, W, Cin)) if need_dx else None
    gw2, gw1, gws = grads.of(blk.conv2.weight), grads.of(blk.conv1.weight), grads.of(blk.shortcut[0].weight)
    n2 = ops.conv_wgrad_slab_count(B, H, W, Cout, Cout, False) if WGRAD == "slabs" else 0
    n1 = ops.conv_wgrad_slab_count(B, H, W, Cin, Cout, True) if WGRAD == "slabs" else 0
    if n1 and n2:
        sl2 = ws.buf(name + ".c2.slabs", (n2, 9, Cout, Cout))
        sl1 = ws.buf(name + ".c1.slabs", (n1, 9, Cin, Cout))
        sls = ws.buf(name + ".sc.slabs", (n1, Cin, Cout))
        if reduce_list is None:
            raise ValueError("block_backward: the slab path needs a reduce list")
        entries = [(sl2, gw2, n2, 9, Cout, Cout), (sl1, gw1, n1, 9, Cin, Cout), (sls, gws, n1, 1, Cin, Cout)]
        if not WGRAD_REDUCE_PER_BLOCK:
            reduce_list += entries

        def wgrad_conv2():                        # needs d_c2 (and the scratch of the elementwise backward)
            ops.resblock_bwd_se_wgrad(scratch, s.hid, s.gap, grads.of(W1), grads.of(W2), B, H, W, Cout)
            ops.conv_wgrad_slabs(s.h1, d_c2, None, sl2, None, B, H, W, Cout, Cout)

        def wgrad_conv1_shortcut():               # needs d_c1 and d_sc; both read the block input
            ops.conv_wgrad_slabs(s.a_in, d_c1, d_sc, sl1, sls, B, H, W, Cin, Cout)
            if WGRAD_REDUCE_PER_BLOCK:
                plan = ws.objs.get(name + ".reduceplan") or ops.ReducePlan()
                if ws.persistent:
                    ws.objs[name + ".reduceplan"] = plan
                plan.run(entries)
    else:
        wsz = max(ops.conv_wgrad_workspace(B, H, W, Cout, Cout, 9), ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 9))
        wg = ws.buf(name + ".wgrad_ws", wsz)
        wgs = ws.buf(name + ".wgrad_ws_sc", ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 1))

        def wgrad_conv2():
            ops.resblock_bwd_se_wgrad(scratch, s.hid, s.gap, grads.of(W1), grads.of(W2), B, H, W, Cout)
            ops.conv_wgrad(s.h1, d_c2, gw2, wg, B, H, W, Cout, Cout, ops.TAPS3X3)

        def wgrad_conv1_shortcut():
            ops.conv_wgrad_pair(s.a_in, d_c1, gw1, d_sc, gws, wg, wgs, B, H, W, Cin, Cout)

    ops.stamp(name + ".main.dc2_ready")
    f2 = fork(ws, d_out.device, enable=FORK_WGRAD).mark()           # d_c2 / d_sc are ready
    # the data gradient of conv2 is the gradient with respect to relu(bn1(c1)): where the strip-sharing kernel runs, its
    # epilogue leaves the two reductions of bn1's backward behind and the separate pass over d_h1 and c1 is not needed
    fused_stats = FUSE_BN_BWD_STATS and ops.conv_shares_tap_rows(B, H, W, Cout, ops.TAPS3X3, negate=True)
    if fused_stats:
        rows, _ = ops.conv_stats_rows(B, H, W, Cout)
        part = ws.buf(name + ".bnrelu_partial", rows * 2 * Cout)
        ops.conv_fwd_bnrelu_bwd_stats(d_c2, s.w2d, d_h1, part, s.c1, s.coeff1, B, H, W, Cout, Cout, ops.TAPS3X3,
                                      negate=True)
    else:
        ops.conv_fwd(d_c2, s.w2d, d_h1, None, B, H, W, Cout, Cout, ops.TAPS3X3, negate=True)
    if deferred is not None:
        deferred.append(wgrad_conv2)
    else:
        with f2:
            ops.stamp(name + ".side.wgrad2_start")
            wgrad_conv2()
            ops.stamp(name + ".side.wgrad2_end")
    if fused_stats:
        ops.bn_relu_bwd_finish(d_h1, s.c1, s.coeff1, part, rows, grads.of(blk.bn1.weight), grads.of(blk.bn1.bias),
                               ws.buf(name + ".bnrelu_cb", 3 * Cout), B * H * W, Cout, training)
    else:
        sc2 = ws.buf(name + ".bnrelu_scratch", ops.bn_relu_bwd_scratch(B * H * W, Cout))
        ops.bn_relu_bwd(d_h1, s.c1, s.coeff1, grads.of(blk.bn1.weight), grads.of(blk.bn1.bias), sc2, B * H * W, Cout,
                        training)
    ops.stamp(name + ".main.dc1_ready")
    f1 = fork(ws, d_out.device, enable=FORK_WGRAD).mark()           # d_c1 is ready
    if need_dx:
        # 3x3 branch and 1x1 shortcut in one launch (one K loop over both operand pairs); on a map with few output
        # tiles the K range is split over two workgroups per tile, which add into the map the forward pass zeroed
        if ws.persistent and ops.conv_dual_prefers_split(B, H, W, Cin):
            ops.conv_fwd_dual_split(d_c1, s.w1d, d_sc, s.wsd, d_in, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
        else:
            ops.conv_fwd_dual(d_c1, s.w1d, d_sc, s.wsd, d_in, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
    if deferred is not None:
        deferred.append(wgrad_conv1_shortcut)
    else:
        with f1:
            ops.stamp(name + ".side.wgrad1_start")
            wgrad_conv1_shortcut()
            ops.stamp(name + ".side.wgrad1_end")
    return d_in


def audio_forward(m, x: torch.Tensor, training: bool, need_grad: bool, ws: Workspace, side_work=None):
    """x: [B,1,H,W] (or [B,H,W]) float32 on the GPU -> (features [B,256], saved state).
    side_work: optional (k, fn): fn() is called once block k's kernels (k = 1..3; 0: the stem's) have been enqueued --
    the caller's independent work (the text branch), which forks its own helper stream there.  fn may return a second
    callable, which is then called behind the first convolution of the following block (so that the critical-path
    kernel is enqueued ahead of the side work that starts at the same point)."""
    if x.dim() == 4:
        if x.shape[1] != 1:
            raise ValueError("audio input must have one channel")
        B, _, H, W = x.shape
    elif x.dim() == 3:
        B, H, W = x.shape
    else:
        raise ValueError("audio input must be [B,1,H,W]")
    if H % 16 or W < 16:
        raise ValueError("spectrogram height must be a multiple of 16 and width at least 16")
    if need_grad and not ops.conv_wgrad_supported(B, H // 2, W // 2, 32, 64):
        raise ValueError(f"a spectrogram of {W} frames is too wide to train on: the weight-gradient kernels stage whole "
                         "rows of the first block's map in LDS (at most 476 frames; inference has no such limit)")
    x = x.contiguous()
    st = SimpleNamespace(B=B, H=H, W=W, x=x, blocks=[])
    conv0, bn0 = m.conv1[0], m.conv1[1]
    g, b, rm, rv, nbt = _bn_args(bn0)
    st.coeff0 = ws.buf("stem.coeff", 128)
    # the weight re-layout depends on nothing in this step: it runs beside the stem (enqueued after it, from a fork
    # point fixed here)
    fpack = fork(ws, x.device).mark()
    partial = None
    srows = ops.stem_stats_rows(B, H)
    if training:
        partial = ws.buf("stem.partial", srows * 64)
        ops.stem_stats(x, conv0.weight, conv0.bias, partial, B, H, W)
    ops.bn_finalize(partial, srows, srows // B, B * H * W, g, b, rm, rv, nbt, training, st.coeff0, None, 32)
    a = ws.buf("stem.out", (B, H // 2, W // 2, 32))
    ops.stem_apply_pool(x, conv0.weight, conv0.bias, st.coeff0, a, B, H, W)
    h, w, cin = H // 2, W // 2, 32
    with fpack:
        packs = pack_all(m, need_grad, ws)
        if need_grad and ws.persistent:
            # input-gradient maps that the backward pass fills by atomic adds of two half-sums start from zero
            hh, ww, cc = h, w, cin
            for i in range(3):
                if ops.conv_dual_prefers_split(B, hh, ww, cc):
                    ws.buf(f"rb{i + 1}.d_in", (B, hh, ww, cc)).zero_()
                hh, ww, cc = hh // 2, ww // 2, cc * 2
    fpack.join()
    ops.stamp("fwd.stem_done")
    pending = None
    if side_work is not None and side_work[0] == 0:
        pending = side_work[1]()
    for i, blk in enumerate((m.res_block1, m.res_block2, m.res_block3)):
        cout = blk.conv1.weight.shape[0]
        feat = ws.buf("audio.feat", (B, cout)) if i == 2 and FUSE_GAP else None
        s = block_forward(blk, f"rb{i + 1}", a, B, h, w, cin, cout, training, need_grad, ws, packs, hook=pending,
                          mean_out=feat)
        pending = None
        st.blocks.append(s)
        a, h, w, cin = s.out, h // 2, w // 2, cout
        ops.stamp(f"fwd.rb{i + 1}_done")
        if side_work is not None and side_work[0] == i + 1:
            pending = side_work[1]()
    if pending is not None:
        pending()
    st.P, st.C = h * w, cin
    if not FUSE_GAP:
        feat = ws.buf("audio.feat", (B, cin))
        ops.gap_fwd(a, feat, B, st.P, cin)
    return feat, st


def audio_backward(m, st, dfeat: torch.Tensor, training: bool, grads: Grads, ws: Workspace):
    B = st.B
    if FUSE_GAP:
        d = dfeat.contiguous()       # the last block's backward takes the gradient of the pooled mean as it is
    else:
        d = ws.buf("audio.d_last", (B, st.P, st.C))
        ops.gap_bwd(dfeat.contiguous(), d, B, st.P, st.C)
    reduce_list = []
    deferred = [] if WGRAD_DEFER and ws.persistent and FORK_WGRAD and FORK_ANY else None
    for i, (blk, s) in enumerate(zip((m.res_block3, m.res_block2, m.res_block1), reversed(st.blocks))):
        d = block_backward(blk, s, d, training, grads, ws, need_dx=True, reduce_list=reduce_list, deferred=deferred,
                           d_out_is_mean=FUSE_GAP and i == 0)
    fw = fork(ws, dfeat.device).mark() if deferred else None     # the last data-gradient convolution is on its way
    conv0, bn0 = m.conv1[0], m.conv1[1]
    scratch = ws.buf("stem.bwd_scratch", ops.stem_bwd_scratch(B, st.H))
    ops.stem_bwd(st.x, conv0.weight, conv0.bias, st.coeff0, d, grads.of(conv0.weight), grads.of(conv0.bias),
                 grads.of(bn0.weight), grads.of(bn0.bias), scratch, B, st.H, st.W, training)
    if deferred:
        with fw:
            for fn in deferred:
                fn()
    if reduce_list:
        # every layer's partial slabs become gradients in one launch, behind the last weight-gradient kernel
        plan = ws.objs.get("audio.reduceplan")
        if plan is None:
            plan = ops.ReducePlan()
            if ws.persistent:
                ws.objs["audio.reduceplan"] = plan
        with fork(ws, dfeat.device, enable=FORK_WGRAD):
            plan.run(reduce_list)
    ops.stamp("audio_bwd.main.chain_end")
    fork(ws, dfeat.device).join()                # all weight gradients are in
    ops.stamp("audio_bwd.main.joined")

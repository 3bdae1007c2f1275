"""Drop-in modules for `CNN implementation/multimodal_model.py`.

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the reference, so
checkpoints written by its training driver (`multimodal_train.py:338-358`) load with ``strict=True`` and a
seeded construction consumes torch's global RNG in the same order (identical initial weights).  The stock
``nn.Conv2d`` / ``nn.Linear`` / ... leaves are used purely as parameter containers: their ``forward`` is never
called.  All arithmetic runs in the HIP library through :mod:`mmsa_amd.audio`, :mod:`mmsa_amd.text` and
:mod:`mmsa_amd.head`; there is no CPU or ATen fallback -- a model left on the CPU raises on ``forward``.

Training support comes in two forms:
* ``loss.backward()`` works (one ``autograd.Function`` spans the whole model; its backward runs the HIP backward
  kernels), so the reference's training loop runs unchanged with ``torch.optim.AdamW`` + ``clip_grad_norm_``;
* :class:`mmsa_amd.trainer.TrainStep` runs forward + backward + clip + AdamW without autograd, over flat
  parameter / gradient buffers, and can replay the whole step from a captured HIP graph.
"""
from __future__ import annotations

import os
import weakref

from typing import List, Optional

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _lib, audio as audio_engine, head as head_engine, ops, text as text_engine, torch_ops  # noqa: F401
from .arena import ParamArena
from .workspace import Grads, Workspace

L2_LAMBDA = 0.001
INFER_CHUNK = 256


def _stack(spec) -> nn.Sequential:
    """Build an ``nn.Sequential`` from a compact spec; positions matter (they are the state_dict indices)."""
    makers = {
        "lin": lambda a, b: nn.Linear(a, b),
        "lin_nobias": lambda a, b: nn.Linear(a, b, bias=False),
        "ln": lambda n: nn.LayerNorm(n),
        "bn1": lambda n: nn.BatchNorm1d(n),
        "bn2": lambda n: nn.BatchNorm2d(n),
        "relu": lambda: nn.ReLU(),
        "gelu": lambda: nn.GELU(),
        "drop": lambda p: nn.Dropout(p),
        "flat": lambda: nn.Flatten(),
        "sigmoid": lambda: nn.Sigmoid(),
        "softmax": lambda: nn.Softmax(dim=1),
        "pool": lambda: nn.MaxPool2d(2, 2),
        "conv3": lambda a, b, bias: nn.Conv2d(a, b, 3, 1, 1, bias=bias),
        "conv1": lambda a, b: nn.Conv2d(a, b, 1, 1, bias=False),
    }
    return nn.Sequential(*[makers[s[0]](*s[1:]) for s in spec])


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b on the HIP dense kernel, with its backward (rows of x: [n, K])."""

    @staticmethod
    def forward(ctx, rows, weight, bias):
        y = torch.empty((rows.shape[0], weight.shape[0]), dtype=torch.float32, device=rows.device)
        ops.dense_fwd(rows, weight, bias, None, None, y, None, None, ops.EPI_NONE)
        ctx.save_for_backward(rows, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        rows, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dpre = torch.empty_like(dy)
        dx = torch.empty_like(rows)
        dW, db = torch.empty_like(weight), torch.empty(weight.shape[0], dtype=torch.float32, device=dy.device)
        ops.dense_bwd(dy, rows, None, None, weight, None, None, dpre, None, dx, dW, db, None, None, ops.EPI_NONE)
        return dx, dW, db


def _channels_last(x: torch.Tensor, dev) -> torch.Tensor:
    """[B, C, H, W] (the reference's layout) -> contiguous [B, H, W, C] float32 on the GPU (the kernels' layout)."""
    if x.dim() != 4:
        raise ValueError("expected a [B, C, H, W] map")
    return x.to(device=dev, dtype=torch.float32).permute(0, 2, 3, 1).contiguous()


class Attention(nn.Module):
    """multimodal_model.py:5-24.  The reference defines it and never instantiates it, so it stays a plain composition:
    the three projections run on the HIP dense kernel, the two small products and the softmax on torch ops."""

    def __init__(self, dim):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.scale = dim ** 0.5

    def forward(self, x):
        dev = _require_gpu(self, "Attention")
        x = x.to(device=dev, dtype=torch.float32)
        rows = x.reshape(-1, x.shape[-1]).contiguous()

        def proj(lin):
            return _LinearFn.apply(rows, lin.weight, lin.bias).reshape(*x.shape[:-1], -1)
        q, k, v = proj(self.query), proj(self.key), proj(self.value)
        attn = torch.softmax(torch.matmul(q, k.transpose(-2, -1)) / self.scale, dim=-1)
        return torch.matmul(attn, v)


class SEBlock(nn.Module):
    """multimodal_model.py:27-42.  Inside the model the block is evaluated by the residual-block kernels; called on
    its own it runs global average -> FC -> ReLU -> FC -> sigmoid -> channel scale as three launches."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = _stack([("lin_nobias", channels, channels // reduction), ("relu",),
                          ("lin_nobias", channels // reduction, channels), ("sigmoid",)])

    def forward(self, x):
        dev = _require_gpu(self, "SEBlock")
        W1, W2 = self.fc[0].weight, self.fc[2].weight
        Ch, C = W1.shape
        if C > 256 or Ch > 16 or C % 4 or Ch < 1:
            raise NotImplementedError("the HIP squeeze-excite kernel takes up to 256 channels and 16 hidden units")
        xl = _channels_last(x, dev)
        if xl.shape[3] != C:
            raise ValueError(f"SEBlock({C}) got {xl.shape[3]} channels")
        return _SEBlockFn.apply(xl, W1, W2).permute(0, 3, 1, 2)


class _SEBlockFn(torch.autograd.Function):
    """SEBlock on its own (channels-last in and out).  Forward: three HIP launches.  Backward: the map-sized work
    (dout * gate) is a HIP launch; the per-utterance [B, C] / [B, C/r] algebra of the two small matrices, a few
    thousand multiply-adds, is left to torch's GPU operators."""

    @staticmethod
    def forward(ctx, xl, W1, W2):
        dev = xl.device
        B, H, W, C = xl.shape
        Ch = W1.shape[0]
        K = torch.ops.mmsa
        mean = torch.empty((B, C), dtype=torch.float32, device=dev)
        K.gap_fwd(xl, mean, B, H * W, C)
        ident = torch.zeros(4 * C, dtype=torch.float32, device=dev)
        ident[C:3 * C] = 1.0                        # [mean 0, 1/std 1, scale 1, shift 0]
        gap, hid, gate = (torch.empty((B, n), dtype=torch.float32, device=dev) for n in (C, Ch, C))
        K.se_gate(mean, ident, W1, W2, 1, gap, hid, gate, B, C)
        out = torch.empty_like(xl)
        K.gate_affine_add(xl, None, gate, None, None, out, B, H * W, C, False)
        ctx.save_for_backward(xl, gap, hid, gate, W1, W2)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        xl, gap, hid, gate, W1, W2 = ctx.saved_tensors
        B, H, W, C = xl.shape
        dout = dout.contiguous()
        dgate = (dout * xl).sum(dim=(1, 2))                      # [B, C]
        du = dgate * gate * (1.0 - gate)                          # through the sigmoid
        dW2 = du.t() @ hid                                        # [C, Ch]
        dhid = (du @ W2) * (hid > 0)                              # through the ReLU
        dW1 = dhid.t() @ gap                                      # [Ch, C]
        dgap = dhid @ W1                                          # [B, C]
        dx = torch.empty_like(xl)
        torch.ops.mmsa.gate_affine_add(dout, None, gate, None, None, dx, B, H * W, C, False)
        dx += (dgap / float(H * W)).view(B, 1, 1, C)
        return dx, dW1, dW2


class ResidualBlock(nn.Module):
    """multimodal_model.py:45-70: conv3x3 -> BN -> ReLU -> conv3x3 -> BN -> SE -> (+ shortcut) -> ReLU, the shortcut
    being the identity when the channel count stays and the stride is 1, else conv1x1 -> BN.  The HIP convolutions are
    stride 1 (the only stride the model uses): a strided block can be built (same parameters) but not run."""

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_channels != out_channels:
            self.shortcut = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride, bias=False),
                                          nn.BatchNorm2d(out_channels))
        self.se = SEBlock(out_channels)
        self.stride = stride

    def forward(self, x):
        dev = _require_gpu(self, "ResidualBlock")
        if self.stride != 1:
            raise NotImplementedError("the HIP convolution kernels are stride 1 (what the model uses)")
        Cout, Cin = self.conv1.weight.shape[:2]
        if Cin % 32 or Cout % 64 or Cout > 256:
            raise NotImplementedError("HIP residual block: in_channels % 32 == 0, out_channels % 64 == 0 and <= 256")
        xl = _channels_last(x, dev)
        if xl.shape[3] != Cin:
            raise ValueError(f"ResidualBlock expects {Cin} channels, got {xl.shape[3]}")
        need_grad = torch.is_grad_enabled() and (xl.requires_grad or any(p.requires_grad for p in self.parameters()))
        if need_grad:
            B, H, W, _ = xl.shape
            if H < 2 or W < 2 or not ops.conv_wgrad_supported(B, H, W, Cin, Cout):
                raise NotImplementedError("ResidualBlock backward: the map is too small or too wide for the "
                                          "weight-gradient kernels (rows of at most about 238 positions)")
        return _ResidualBlockFn.apply(self, xl, need_grad, *self.parameters()).permute(0, 3, 1, 2)


class _ResidualBlockFn(torch.autograd.Function):
    """ResidualBlock on its own (channels-last in and out): the launch sequence the audio branch uses for a block, minus
    the pooling that follows a block there; forward and backward are HIP launches throughout."""

    @staticmethod
    def forward(ctx, m, xl, need_grad, *params):
        dev = xl.device
        B, H, W, Cin = xl.shape
        Cout = m.conv1.weight.shape[0]
        K = torch.ops.mmsa
        training = audio_engine.block_bn_training(m)
        project = len(m.shortcut) > 0
        npos = B * H * W
        packs = {}

        def conv_bn(tag, conv, bn, X, cin, taps, want_bsum):
            """conv (+ statistics epilogue) and the BatchNorm coefficient kernel; returns (raw map, coeff, bsum)."""
            wp = torch.empty(conv.weight.numel(), dtype=torch.float32, device=dev)
            wd = torch.empty(conv.weight.numel(), dtype=torch.float32, device=dev) if need_grad else None
            K.conv_pack_weights(conv.weight, wp, wd)
            packs[tag] = wd
            Y = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
            rows, tiles = ops.conv_stats_rows(B, H, W, Cout)
            stats = torch.empty(rows * 2 * Cout, dtype=torch.float32, device=dev) if (training or want_bsum) else None
            K.conv_fwd(X, wp, Y, stats, B, H, W, cin, Cout, [t[0] for t in taps], [t[1] for t in taps])
            coeff = torch.empty(4 * Cout, dtype=torch.float32, device=dev)
            bsum = torch.empty((B, Cout), dtype=torch.float32, device=dev) if want_bsum else None
            K.bn_finalize(stats, rows, tiles, npos, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                          bn.num_batches_tracked, training, coeff, bsum, Cout)
            return Y, coeff, bsum

        c1, coeff1, _ = conv_bn("c1", m.conv1, m.bn1, xl, Cin, ops.TAPS3X3, False)
        h1 = torch.empty_like(c1)
        K.bn_relu_apply(c1, coeff1, h1, npos, Cout)
        c2, coeff2, bsum2 = conv_bn("c2", m.conv2, m.bn2, h1, Cout, ops.TAPS3X3, True)
        W1, W2 = m.se.fc[0].weight, m.se.fc[2].weight
        gap, hid, gate = (torch.empty((B, n), dtype=torch.float32, device=dev) for n in (Cout, W1.shape[0], Cout))
        K.se_gate(bsum2, coeff2, W1, W2, H * W, gap, hid, gate, B, Cout)
        if project:
            sc, coeffs, _ = conv_bn("sc", m.shortcut[0], m.shortcut[1], xl, Cin, ops.TAPS1X1, False)
        else:
            sc, coeffs = xl, None
        out = torch.empty((B, H, W, Cout), dtype=torch.float32, device=dev)
        K.gate_affine_add(c2, coeff2, gate, sc, coeffs, out, B, H * W, Cout, True)
        if need_grad:
            ctx.m = m
            ctx.st = (xl, c1, coeff1, h1, c2, coeff2, bsum2, gap, hid, gate, sc, coeffs, packs, training, project)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        m = ctx.m
        xl, c1, coeff1, h1, c2, coeff2, bsum2, gap, hid, gate, sc, coeffs, packs, training, project = ctx.st
        dev = xl.device
        B, H, W, Cin = xl.shape
        Cout = c2.shape[3]
        npos = B * H * W
        f32 = dict(dtype=torch.float32, device=dev)
        grads = Grads()
        W1, W2 = m.se.fc[0].weight, m.se.fc[2].weight
        if project:
            bns = m.shortcut[1]
            dgs, dbs = grads.of(bns.weight), grads.of(bns.bias)
            flags = int(training)
        else:
            # identity shortcut: a fixed affine map with scale 1 (multimodal_model.py:56-61 leaves the Sequential empty)
            coeffs = torch.zeros(4 * Cout, **f32)
            coeffs[Cout:3 * Cout] = 1.0
            dgs, dbs = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
            flags = int(training) | 2
        d_c2 = torch.empty((B, H, W, Cout), **f32)
        d_sc = torch.empty((B, H, W, Cout), **f32)
        scratch = torch.empty(ops.resblock_bwd_scratch(B, H, W, Cout, unpooled=True), **f32)
        ops.resblock_bwd_ew(dout.contiguous(), c2, sc, coeff2, coeffs, gate, hid, gap, bsum2, W1, W2, d_c2, d_sc,
                            grads.of(m.bn2.weight), grads.of(m.bn2.bias), dgs, dbs, grads.of(W1), grads.of(W2), scratch,
                            B, H, W, Cout, flags, unpooled=True)
        wsz = max(ops.conv_wgrad_workspace(B, H, W, Cout, Cout, 9), ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 9),
                  ops.conv_wgrad_workspace(B, H, W, Cin, Cout, 1))
        wbuf = torch.empty(wsz, **f32)
        # second convolution: data gradient (flipped taps), weight gradient, then through ReLU and bn1 in place
        d_h1 = torch.empty((B, H, W, Cout), **f32)
        ops.conv_fwd(d_c2, packs["c2"], d_h1, None, B, H, W, Cout, Cout, ops.TAPS3X3, negate=True)
        ops.conv_wgrad(h1, d_c2, grads.of(m.conv2.weight), wbuf, B, H, W, Cout, Cout, ops.TAPS3X3)
        ops.bn_relu_bwd(d_h1, c1, coeff1, grads.of(m.bn1.weight), grads.of(m.bn1.bias),
                        torch.empty(ops.bn_relu_bwd_scratch(npos, Cout), **f32), npos, Cout, training)
        d_c1 = d_h1
        # first convolution and the shortcut
        d_in = torch.empty((B, H, W, Cin), **f32)
        if project:
            ops.conv_fwd_dual(d_c1, packs["c1"], d_sc, packs["sc"], d_in, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
            ops.conv_wgrad(xl, d_sc, grads.of(m.shortcut[0].weight), wbuf, B, H, W, Cin, Cout, ops.TAPS1X1)
        else:
            ops.conv_fwd(d_c1, packs["c1"], d_in, None, B, H, W, Cout, Cin, ops.TAPS3X3, negate=True)
            d_in += d_sc
        ops.conv_wgrad(xl, d_c1, grads.of(m.conv1.weight), wbuf, B, H, W, Cin, Cout, ops.TAPS3X3)
        return (None, d_in, None) + tuple(grads.of(p, zero=True) for p in m.parameters())


def _require_gpu(module: nn.Module, what: str):
    p = next(module.parameters())
    if not p.is_cuda:
        raise _lib.MmsaError(f"{what}: parameters are on {p.device}; the HIP kernels run on an MI355X only "
                             "(move the module with .to('cuda')); there is no CPU fallback")
    _lib.load()
    return p.device


def _needs_grad(module: nn.Module) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


def _refuse_spectrogram_grad(x) -> None:
    """The stem's backward forms parameter gradients only (training never asks for more): a gradient with respect to
    the spectrogram would silently come back as None, so asking for one is refused."""
    if x is not None and torch.is_grad_enabled() and x.requires_grad:
        raise NotImplementedError("gradients with respect to the input spectrogram are not computed by the HIP "
                                  "backward pass; detach the input")


class _BranchFn(torch.autograd.Function):
    """Autograd bridge for a single branch used on its own (AudioModel(x) / TextModel(ids))."""

    @staticmethod
    def forward(ctx, module, inp, *params):
        ws = Workspace(inp.device, persistent=False)
        training = module.training
        if isinstance(module, AudioModel):
            training = audio_engine.bn_training(module)
            feat, st = audio_engine.audio_forward(module, inp, training, True, ws)
        else:
            feat, st = text_engine.text_forward(module, inp, True, ws)
            text_engine.check_ids(st.err)
        ctx.module, ctx.st, ctx.ws, ctx.training = module, st, ws, training
        ctx.set_materialize_grads(False)
        return feat.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, dfeat):
        m = ctx.module
        plist = list(m.parameters())
        if dfeat is None:
            return (None, None) + tuple(None for _ in plist)
        grads = Grads()
        if isinstance(m, AudioModel):
            audio_engine.audio_backward(m, ctx.st, dfeat, ctx.training, grads, ctx.ws)
        else:
            text_engine.text_backward(m, ctx.st, dfeat, grads, ctx.ws)
        return (None, None) + tuple(grads.get(p) for p in plist)


class AudioModel(nn.Module):
    def __init__(self, num_classes=3, dropout_rate=0.5):
        super().__init__()
        self.conv1 = _stack([("conv3", 1, 32, True), ("bn2", 32), ("relu",), ("pool",)])
        self.res_block1 = ResidualBlock(32, 64)
        self.pool1 = nn.MaxPool2d(2, 2)
        self.res_block2 = ResidualBlock(64, 128)
        self.pool2 = nn.MaxPool2d(2, 2)
        self.res_block3 = ResidualBlock(128, 256)
        self.pool3 = nn.MaxPool2d(2, 2)
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.feature_dim = 256
        self._initialize_weights()

    def _initialize_weights(self):
        # multimodal_model.py:104-116: He-normal (fan_out) convs, unit BatchNorm, N(0, 0.01) SE matrices
        for mod in self.modules():
            if isinstance(mod, nn.Conv2d):
                nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
                if mod.bias is not None:
                    nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.Linear):
                nn.init.normal_(mod.weight, 0, 0.01)

    def forward(self, x):
        dev = _require_gpu(self, "AudioModel")
        _refuse_spectrogram_grad(x)
        x = x.to(device=dev, dtype=torch.float32)
        if _needs_grad(self):
            return _BranchFn.apply(self, x, *self.parameters())
        ws = self.__dict__.setdefault("_ws", Workspace(dev))
        feat, _ = audio_engine.audio_forward(self, x, audio_engine.bn_training(self), False, ws)
        return feat.clone()


class TextModel(nn.Module):
    def __init__(self, vocab_size=30522, embedding_dim=256, filter_sizes=[3, 4, 5], num_filters=128,
                 dropout_rate=0.3, max_length=128):
        super().__init__()
        self.max_length = max_length
        self.embedding_dim = embedding_dim
        self.embedding = nn.Embedding(vocab_size, embedding_dim, padding_idx=0)
        self.convs = nn.ModuleList([nn.Conv1d(embedding_dim, num_filters, kernel_size=k) for k in filter_sizes])
        self.feature_dim = len(filter_sizes) * num_filters
        self._init_weights()

    def _init_weights(self):
        # multimodal_model.py:163-171 (the normal_ draw also overwrites the padding row)
        nn.init.normal_(self.embedding.weight, mean=0, std=0.1)
        for conv in self.convs:
            nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
            nn.init.constant_(conv.bias, 0)

    def forward(self, input_ids, attention_mask=None):
        dev = _require_gpu(self, "TextModel")
        input_ids = input_ids.to(dev)
        if _needs_grad(self):
            return _BranchFn.apply(self, input_ids, *self.parameters())
        ws = self.__dict__.setdefault("_ws", Workspace(dev))
        feat, st = text_engine.text_forward(self, input_ids, False, ws)
        text_engine.check_ids(st.err)
        return feat.clone()


class MultifactorFeatureNormalizer(nn.Module):
    def __init__(self, audio_dim, text_dim):
        super().__init__()
        self.audio_norm = nn.LayerNorm(audio_dim)
        self.text_norm = nn.LayerNorm(text_dim)
        self.audio_scale = nn.Parameter(torch.ones(1))
        self.text_scale = nn.Parameter(torch.ones(1))

    def forward(self, audio_features, text_features):
        """multimodal_model.py:216-225: (LayerNorm(audio) * audio_scale, LayerNorm(text) * text_scale)."""
        dev = _require_gpu(self, "MultifactorFeatureNormalizer")
        xs = []
        for x, norm in ((audio_features, self.audio_norm), (text_features, self.text_norm)):
            x = x.to(device=dev, dtype=torch.float32).contiguous()
            if x.dim() != 2 or x.shape[1] != norm.weight.shape[0]:
                raise ValueError("features must be [B, dim] with the dims the normaliser was built for")
            xs.append(x)
        return _NormalizerFn.apply(self, xs[0], xs[1], *self.parameters())


class _NormalizerFn(torch.autograd.Function):
    """The feature normaliser on its own: two `ln_scale` launches forward, two backward."""

    @staticmethod
    def forward(ctx, m, a, t, *params):
        outs, saved = [], []
        for x, norm, scale in ((a, m.audio_norm, m.audio_scale), (t, m.text_norm, m.text_scale)):
            y = torch.empty_like(x)
            stats = torch.empty(2 * x.shape[0], dtype=torch.float32, device=x.device)
            ops.ln_scale_fwd(x, norm.weight, norm.bias, scale, None, y, stats)
            outs.append(y)
            saved += [x, stats]
        ctx.m = m
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, dya, dyt):
        m = ctx.m
        xa, sa, xt, st = ctx.saved_tensors
        grads = {}
        dxs = []
        for dy, x, stats, norm, scale in ((dya, xa, sa, m.audio_norm, m.audio_scale),
                                          (dyt, xt, st, m.text_norm, m.text_scale)):
            B, N = x.shape
            dy = torch.zeros_like(x) if dy is None else dy.contiguous()
            dx = torch.empty_like(x)
            dg, db, ds = torch.empty_like(norm.weight), torch.empty_like(norm.bias), torch.empty_like(scale)
            ops.ln_scale_bwd(dy, x, norm.weight, norm.bias, scale, None, stats, dx, dg, db, ds,
                             torch.empty(B * N + B, dtype=torch.float32, device=x.device))
            dxs.append(dx)
            grads[id(norm.weight)], grads[id(norm.bias)], grads[id(scale)] = dg, db, ds
        return (None, dxs[0], dxs[1]) + tuple(grads[id(p)] for p in m.parameters())


class ModalityFusion(nn.Module):
    def __init__(self, audio_dim, text_dim, hidden_dim=256, dropout_rate=0.5, audio_dropout_prob=0.3,
                 text_dropout_prob=0.3):
        super().__init__()
        self.audio_projector = _stack([("lin", audio_dim, hidden_dim), ("ln", hidden_dim), ("relu",),
                                       ("drop", dropout_rate)])
        self.text_projector = _stack([("lin", text_dim, hidden_dim), ("ln", hidden_dim), ("relu",),
                                      ("drop", dropout_rate)])
        self.attention = _stack([("lin", hidden_dim * 2, 2), ("softmax",)])
        self.output_dim = hidden_dim
        self.audio_dropout_prob = audio_dropout_prob
        self.text_dropout_prob = text_dropout_prob

    def draw_modality_dropout(self):
        """The reference's two host-side draws, audio first (multimodal_model.py:264-268)."""
        if not self.training:
            return False, False
        da = torch.rand(1).item() < self.audio_dropout_prob
        dt = torch.rand(1).item() < self.text_dropout_prob
        return da, dt

    def forward(self, audio_features, text_features):
        """multimodal_model.py:263-285: modality dropout (two host draws, training only), the two projectors
        (Linear -> LayerNorm -> ReLU -> Dropout), the two-way attention softmax and the weighted sum."""
        dev = _require_gpu(self, "ModalityFusion")
        a = audio_features.to(device=dev, dtype=torch.float32).contiguous()
        t = text_features.to(device=dev, dtype=torch.float32).contiguous()
        da, dt = self.draw_modality_dropout()
        p = self.audio_projector[3].p if self.training else 0.0
        seed = (torch.initial_seed() * 1000003 + int(torch.randint(0, 2 ** 31, (1,)).item())) & ((1 << 62) - 1) if p > 0 else 0
        return _FusionFn.apply(self, a, t, bool(da), bool(dt), float(p), int(seed), *self.parameters())


class _FusionFn(torch.autograd.Function):
    """The fusion layer on its own: the launch sequence of the layer-by-layer head (dense kernels with the
    LayerNorm + ReLU + dropout epilogue, the two-way softmax mix) and its backward."""

    @staticmethod
    def forward(ctx, m, a, t, drop_a, drop_t, p, seed, *params):
        from .head import _Dense
        dev = a.device
        ws = Workspace(dev, persistent=False)
        if drop_a:
            a = torch.zeros_like(a)
        if drop_t:
            t = torch.zeros_like(t)
        B, Hd = a.shape[0], m.output_dim
        P = torch.empty((B, 2 * Hd), dtype=torch.float32, device=dev)
        pa = _Dense(ws, "pa", m.audio_projector[0], m.audio_projector[1], ops.EPI_LN_RELU, B, p, seed)
        pt = _Dense(ws, "pt", m.text_projector[0], m.text_projector[1], ops.EPI_LN_RELU, B, p, seed + 1)
        pa.fwd(a, P[:, :Hd])
        pt.fwd(t, P[:, Hd:])
        att = _Dense(ws, "att", m.attention[0], None, ops.EPI_NONE, B)
        att_out = att.fwd(P, torch.empty((B, 2), dtype=torch.float32, device=dev))
        wts = torch.empty(2 * B, dtype=torch.float32, device=dev)
        fused = torch.empty((B, Hd), dtype=torch.float32, device=dev)
        ops.fusion_mix_fwd(P, att_out, wts, fused)
        ctx.m, ctx.st = m, (ws, pa, pt, att, P, wts, drop_a, drop_t)
        return fused

    @staticmethod
    @once_differentiable
    def backward(ctx, dfused):
        m = ctx.m
        ws, pa, pt, att, P, wts, drop_a, drop_t = ctx.st
        B, Hd = P.shape[0], m.output_dim
        dev = P.device
        grads = Grads()
        dP = torch.empty_like(P)
        datt = torch.empty((B, 2), dtype=torch.float32, device=dev)
        ops.fusion_mix_bwd(dfused.contiguous(), P, wts, dP, datt)
        att.bwd(ws, datt, dP, grads, accumulate_dx=True)
        d_a = torch.empty((B, pa.K), dtype=torch.float32, device=dev)
        d_t = torch.empty((B, pt.K), dtype=torch.float32, device=dev)
        pa.bwd(ws, dP[:, :Hd], d_a, grads)
        pt.bwd(ws, dP[:, Hd:], d_t, grads)
        # a dropped modality's features were replaced by zeros (multimodal_model.py:269-273): no gradient reaches them
        if drop_a:
            d_a.zero_()
        if drop_t:
            d_t.zero_()
        return (None, d_a, d_t, None, None, None, None) + tuple(grads.of(q, zero=True) for q in m.parameters())


# per model: one int of pinned host memory and an event, through which the eager forward learns whether the embedding
# gather met a token id outside the table (kept outside the module: neither copies nor pickles well)
_ID_PROBES = weakref.WeakKeyDictionary()


def _ids_probe(model, dev):
    probe = _ID_PROBES.get(model)
    if probe is None or probe[2] != dev:
        probe = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(), dev)
        _ID_PROBES[model] = probe
    return probe


class _ModelFn(torch.autograd.Function):
    """One autograd node for the whole model: forward and backward are HIP launch sequences."""

    @staticmethod
    def forward(ctx, model, audio, input_ids, labels, drops, *params):
        ws = Workspace(params[0].device, False)
        ws.objs["text.err_probe"] = _ids_probe(model, params[0].device)
        run = model._run(audio, input_ids, labels, drops, need_grad=True, ws=ws)
        if run.text is not None or run.err is not None:
            # raises IndexError like the reference's nn.Embedding / nn.CrossEntropyLoss, having waited for the head of
            # the pass (label check, gather), not for the pass
            text_engine.check_ids_probe(ws.objs["text.err_probe"])
        ctx.model, ctx.run, ctx.arena = model, run, model._arena
        ctx.set_materialize_grads(False)
        loss = run.loss.view(()).clone() if run.loss is not None else None
        return run.head.logits.clone(), run.head.probs.clone(), loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g_logits, g_probs, g_loss):
        model, run, arena = ctx.model, ctx.run, ctx.arena
        plist = arena.params
        d = None
        if g_loss is not None and run.head.dlogits is not None:
            d = run.head.dlogits * g_loss
        if g_logits is not None:
            d = g_logits if d is None else d + g_logits
        if g_probs is not None:
            p = run.head.probs
            dp = p * (g_probs - (g_probs * p).sum(dim=1, keepdim=True))
            d = dp if d is None else d + dp
        none = (None,) * 5
        if d is None:
            return none + tuple(None for _ in plist)
        gflat = torch.empty_like(arena.flat)
        views = arena.views_of(gflat)
        grads = Grads({id(p): v for p, v in zip(plist, views)})
        model._backward(run, d.contiguous(), grads)
        if run.l2 and g_loss is not None:
            # d/dp of 0.001 * ||p||: lambda * p / ||p|| per tensor, scaled by the incoming loss gradient
            # (the incoming gradient stays on the device: reading it here would make the host wait for the whole
            # backward pass it has just enqueued)
            ops.l2_grad_add(gflat, arena.flat, arena.tnorm, L2_LAMBDA, arena.table,
                            scale=g_loss.detach().to(torch.float32).reshape(1))
        missing = [p for p in plist if id(p) not in run.touched]
        for p in missing:
            grads.of(p).zero_()
        return none + tuple(views)


# where the text branch's helper stream forks off the audio branch: -1 ahead of the stem, 0 behind the stem (default),
# k = 1..3 behind residual block k (A/B aid)
_TEXT_FORK_AT = int(os.environ.get("MMSA_TEXT_FORK_AT", "0"))
# 1: the fork point is fixed there, but the text branch's kernels are enqueued behind the next block's first convolution
_TEXT_FORK_MARK = os.environ.get("MMSA_TEXT_FORK_MARK", "0") == "1"


class MultimodalSentimentModel(nn.Module):
    def __init__(self, modality="both", num_classes=3, audio_dropout=0.5, text_dropout=0.3, fusion_dropout=0.5,
                 vocab_size=30522, embedding_dim=256, filter_sizes=[3, 4, 5], num_filters=128, max_length=128):
        super().__init__()
        if modality not in ("audio", "text", "both"):
            modality = "both" if modality is None else modality
        self.modality = modality
        self._dropouts = (audio_dropout, text_dropout, fusion_dropout)
        if modality in ("audio", "both"):
            self.audio_model = AudioModel(num_classes=num_classes, dropout_rate=audio_dropout)
            self.audio_feature_dim = self.audio_model.feature_dim
        else:
            self.audio_model = None
            self.audio_feature_dim = 0
        if modality in ("text", "both"):
            self.text_model = TextModel(vocab_size=vocab_size, embedding_dim=embedding_dim, filter_sizes=filter_sizes,
                                        num_filters=num_filters, dropout_rate=text_dropout, max_length=max_length)
            self.text_feature_dim = self.text_model.feature_dim
        else:
            self.text_model = None
            self.text_feature_dim = 0
        if modality == "both":
            self.feature_normalizer = MultifactorFeatureNormalizer(self.audio_feature_dim, self.text_feature_dim)
            self.fusion_layer = ModalityFusion(self.audio_feature_dim, self.text_feature_dim, hidden_dim=256,
                                               dropout_rate=fusion_dropout)
            self.fusion_dim = self.fusion_layer.output_dim
        else:
            self.fusion_dim = self.audio_feature_dim + self.text_feature_dim
        d = self.fusion_dim
        if modality == "audio":
            p = audio_dropout
            spec = [("flat",), ("lin", d, 128), ("bn1", 128), ("relu",), ("drop", p), ("lin", 128, 64), ("bn1", 64),
                    ("relu",), ("drop", p / 2), ("lin", 64, num_classes)]
        elif modality == "text":
            p = text_dropout
            spec = [("drop", p), ("lin", d, 256), ("relu",), ("ln", 256), ("drop", p), ("lin", 256, 128), ("relu",),
                    ("ln", 128), ("drop", p), ("lin", 128, num_classes)]
        else:
            p = fusion_dropout
            spec = [("lin", d, 128), ("ln", 128), ("relu",), ("drop", p), ("lin", 128, 64), ("ln", 64), ("relu",),
                    ("drop", p / 2), ("lin", 64, num_classes)]
        self.classifier = _stack(spec)
        self.loss_fn = nn.CrossEntropyLoss()
        self._arena: Optional[ParamArena] = None
        self._seed_counter = 0

    # ------------------------------------------------------------------ plumbing
    def _ensure_arena(self) -> ParamArena:
        """The flat parameter storage, rebuilt when the parameters no longer live in it: after ``.to()`` /
        ``.float()`` (new storage) or after a layer was exchanged (other Parameter objects).  One walk over the module
        tree per call; callers take the parameter list from the arena instead of walking again."""
        dev = _require_gpu(self, "MultimodalSentimentModel")
        a = self.__dict__.get("_arena")
        cur = list(self.parameters())
        if (a is None or a.device != dev or len(cur) != len(a.params)
                or any(x is not y for x, y in zip(cur, a.params)) or not a.owns()):
            for p in cur:
                if p.dtype != torch.float32:
                    raise TypeError("the HIP kernels compute in float32; convert the model with .float()")
            a = ParamArena(cur, dev)
            self.__dict__["_arena"] = a
            self.__dict__["_ws_infer"] = Workspace(dev)
            self.__dict__["_consts"] = {1: torch.ones(1, device=dev), 0: torch.zeros(1, device=dev)}
        return a

    def _drop_p(self):
        return {"audio": self._dropouts[0], "text": self._dropouts[1], "both": self._dropouts[2]}[self.modality]

    def _next_seed(self):
        self.__dict__["_seed_counter"] = self.__dict__.get("_seed_counter", 0) + 1
        return (torch.initial_seed() * 1000003 + self._seed_counter * 97) & ((1 << 62) - 1)

    def _check_inputs(self, audio, input_ids, labels=None, audio_dims=(3, 4)):
        """The reference's three ValueErrors (multimodal_model.py:419-435), and what its torch layers would complain
        about further down: the kernels take the batch size from one input and trust the others to match, so rows
        that are not there must be refused here, on the host."""
        if self.modality == "audio" and audio is None:
            raise ValueError("Audio input is required for audio modality")
        if self.modality == "text" and input_ids is None:
            raise ValueError("Text input is required for text modality")
        if self.modality == "both" and (audio is None or input_ids is None):
            raise ValueError("Both audio and text inputs are required for multimodal fusion")
        B = None
        if self.modality != "text":
            if audio.dim() not in audio_dims:
                raise ValueError(f"audio input has {audio.dim()} dimensions, expected {' or '.join(map(str, audio_dims))}")
            B = audio.shape[0]
        if self.modality != "audio":
            if input_ids.dim() != 2:
                raise ValueError("input_ids must be [batch, tokens]")
            if B is not None and input_ids.shape[0] != B:
                raise ValueError(f"audio holds {B} utterances, input_ids {input_ids.shape[0]}")
            B = input_ids.shape[0]
        if B == 0:
            raise ValueError("empty batch")
        if labels is not None and (labels.dim() != 1 or labels.shape[0] != B):
            raise ValueError(f"labels must be [{B}] (one class index per utterance), got {tuple(labels.shape)}")

    def _run(self, audio, input_ids, labels, drops, need_grad, ws, keep=None, logits_out=None, norms_ready=False,
             grad_scale=1.0, step_state=None):
        """Forward launch sequence.  Returns a record with the head state, branch states and the loss buffer."""
        from types import SimpleNamespace
        arena = self._arena
        training = self.training
        run = SimpleNamespace(audio=None, text=None, l2=False, touched=set())
        # dropout is applied inside fused kernels that take one rate: the Dropout modules cannot go their own way
        groups = [self.classifier]
        if self.modality == "both":
            groups += [self.fusion_layer.audio_projector, self.fusion_layer.text_projector]
        for seq in groups:
            for mod in seq:
                if isinstance(mod, nn.Dropout) and mod.training != training:
                    raise NotImplementedError("a Dropout module whose training flag differs from the model's (dropout "
                                              "at inference, or a single layer switched off): the HIP head applies "
                                              "dropout by the model's flag")
        # the branch's BatchNorm layers decide by their own flags (kept for the backward pass of this forward pass)
        run.audio_training = audio_engine.bn_training(self.audio_model) if self.audio_model is not None else False
        a_feat = t_feat = None
        # A class label outside [0, C) is an IndexError in the reference (nn.CrossEntropyLoss); the loss kernels would
        # treat such a row as one without a target.  One small launch at the head of the pass -- for 'both' on the text
        # branch's stream -- raises bit 1 of the sticky flag word whose bit 0 belongs to the token ids; the callers
        # turn the flag into the error.
        run.err = None
        if labels is not None:
            labels = labels.contiguous()
            if labels.dtype != torch.int64:
                labels = labels.to(torch.int64)
            run.err = ws.buf("text.err", 1, torch.int32, zero=not ws.persistent)
            ncls = self.classifier[len(self.classifier) - 1].weight.shape[0]

        def label_check():
            if run.err is not None:
                ops.check_labels(labels, ncls, run.err)
        if self.modality == "both":
            # the two branches are independent until the head: the (small) text branch runs on a helper stream; it is
            # enqueued first so that it starts beside the stem instead of colliding with the last block's convolutions
            where = _TEXT_FORK_AT
            box = {}

            def text_branch():
                fk = audio_engine.fork(ws, arena.device, 1)
                box["fork"] = fk

                def body():
                    with fk:
                        label_check()
                        box["out"] = text_engine.text_forward(self.text_model, input_ids, need_grad, ws, err=run.err)
                if _TEXT_FORK_MARK and where >= 0:
                    fk.mark()
                    return body
                body()
                return None
            if where < 0:
                text_branch()
            a_feat, run.audio = audio_engine.audio_forward(self.audio_model, audio, run.audio_training,
                                                           need_grad, ws, side_work=(where, text_branch) if where >= 0
                                                           else None)
            t_feat, run.text = box["out"]
            box["fork"].join()
        elif self.modality == "audio":
            if run.err is not None:
                label_check()
                # (no text branch to carry the flag to the host: the copies it would make)
                probe, host = ws.objs.get("text.err_probe"), ws.objs.get("text.err_host")
                if probe is not None:
                    probe[0].copy_(run.err, non_blocking=True)
                    probe[1].record()
                if host is not None:
                    host.copy_(run.err, non_blocking=True)
            a_feat, run.audio = audio_engine.audio_forward(self.audio_model, audio, run.audio_training,
                                                           need_grad, ws)
        else:
            label_check()
            t_feat, run.text = text_engine.text_forward(self.text_model, input_ids, need_grad, ws, err=run.err)
        seed = self._next_seed() if training and self._drop_p() > 0 else 0
        if self.modality == "both":
            if keep is None:
                c = self._consts
                keep = (c[0] if drops[0] else c[1], c[0] if drops[1] else c[1])
            run.head = head_engine.head_forward_both(self, a_feat, t_feat, labels, training, keep[0], keep[1],
                                                     self._drop_p(), seed, need_grad, ws, logits_out, grad_scale, step_state)
        elif self.modality == "audio":
            run.head = head_engine.head_forward_audio(self, a_feat, labels, training, self._drop_p(), seed, need_grad,
                                                      ws, logits_out, step_state)
        else:
            run.head = head_engine.head_forward_text(self, t_feat, labels, training, self._drop_p(), seed, need_grad,
                                                     ws, logits_out, step_state)
        run.loss = run.head.loss
        if labels is not None and self.modality == "both" and run.head.loss is not None:
            run.l2 = True
            psum = ws.buf("loss.psum", 1)
            run.loss = ws.buf("loss.total", 1)
            # the loss VALUE feeds nothing in the step (the regulariser's gradient is formed from the per-tensor norms
            # inside clip / AdamW): with a gradient wanted it is computed beside the backward, on the helper stream
            # that also carries the head's cross-entropy value and parameter gradients
            head_loss = run.head.loss

            def total_loss():
                ops.param_norms(arena.flat, arena.table, arena.tnorm, psum, norms_ready)
                ops.scalar_axpy(run.loss, head_loss, psum, L2_LAMBDA)
            if need_grad and ws.persistent:
                run.head.deferred = getattr(run.head, "deferred", []) + [total_loss]
            else:
                total_loss()
        if need_grad:
            run.touched = {id(p) for p in arena.params}
        return run

    def _backward(self, run, dlogits, grads: Grads, ws: Optional[Workspace] = None, phase: str = "all"):
        """``phase`` splits the 'both' backward for data-parallel runs: "early" = head + text branch (their gradients,
        87 % of the bytes, can be exchanged while the rest runs), "late" = audio branch."""
        ws = ws or Workspace(dlogits.device, persistent=False)
        if phase != "all" and self.modality != "both":
            raise ValueError("the split backward exists for modality 'both' only")
        dev = dlogits.device if dlogits is not None else self._arena.device

        def side_work():
            """Everything that only the optimiser or the caller reads: the head's parameter gradients, then the loss
            values, on helper stream 2."""
            for fn in getattr(run.head, "side_work", []):
                fn()
            deferred = getattr(run.head, "deferred", [])
            if deferred:
                with audio_engine.fork(ws, dev, 2):
                    for fn in deferred:
                        fn()
            run.head.side_work, run.head.deferred = [], []

        if phase == "early":
            run.d_audio, d_t = head_engine.head_backward_both(self, run.head, dlogits, grads, ws)
            text_engine.text_backward(self.text_model, run.text, d_t, grads, ws)
            side_work()
            audio_engine.fork(ws, dev, 2).join()
            return
        if phase == "late":
            audio_engine.audio_backward(self.audio_model, run.audio, run.d_audio, run.audio_training, grads, ws)
            return
        if self.modality == "both":
            d_a, d_t = head_engine.head_backward_both(self, run.head, dlogits, grads, ws)
            ops.stamp("bwd.head_done")
            # launch order matters under graph replay: the short text branch and the side work are enqueued first (they
            # then start at once on helper queues), the audio chain after them
            with audio_engine.fork(ws, dev, 1) as f:
                text_engine.text_backward(self.text_model, run.text, d_t, grads, ws)
            side_work()
            audio_engine.audio_backward(self.audio_model, run.audio, d_a, run.audio_training, grads, ws)
            f.join()
        elif self.modality == "audio":
            d_a = head_engine.head_backward_audio(self, run.head, dlogits, grads, ws)
            audio_engine.audio_backward(self.audio_model, run.audio, d_a, run.audio_training, grads, ws)
            side_work()
        else:
            d_t = head_engine.head_backward_text(self, run.head, dlogits, grads, ws)
            text_engine.text_backward(self.text_model, run.text, d_t, grads, ws)
            side_work()
        audio_engine.fork(ws, dlogits.device, 2).join()      # the head's parameter gradients (helper stream 2)

    # ------------------------------------------------------------------ the reference's forward signature
    def forward(self, audio=None, input_ids=None, attention_mask=None, labels=None):
        self._check_inputs(audio, input_ids, labels)
        arena = self._ensure_arena()
        # kernels are launched on the current device: make that the model's for the duration of the call (a model moved
        # to cuda:1 in a process that never called torch.cuda.set_device)
        with torch.cuda.device(arena.device):
            return self._forward_on_device(arena, audio, input_ids, labels)

    def _forward_on_device(self, arena, audio, input_ids, labels):
        dev = arena.device
        # the model runs as one launch sequence: the sub-modules' own forward methods are not called, so hooks
        # registered on them would silently never fire
        for child in (self.audio_model, self.text_model, self.__dict__["_modules"].get("feature_normalizer"),
                      self.__dict__["_modules"].get("fusion_layer"), self.classifier):
            if child is not None and (child._forward_hooks or child._forward_pre_hooks or child._backward_hooks
                                      or child._backward_pre_hooks):
                raise NotImplementedError("hooks on a sub-module of MultimodalSentimentModel do not fire (the model runs "
                                          "as one fused launch sequence): hook the model itself, or call the "
                                          "sub-module on its own -- each has a stand-alone forward")
        if audio is not None and self.modality != "text":
            _refuse_spectrogram_grad(audio)
            audio = audio.to(device=dev, dtype=torch.float32)
        if input_ids is not None and self.modality != "audio":
            input_ids = input_ids.to(dev)
        if labels is not None:
            labels = labels.to(dev)
        drops = (False, False)
        if self.modality == "both":
            drops = self.fusion_layer.draw_modality_dropout()
        if torch.is_grad_enabled() and any(p.requires_grad for p in arena.params):
            logits, probs, loss = _ModelFn.apply(self, audio, input_ids, labels, drops, *arena.params)
            return {"loss": loss, "logits": logits, "probabilities": probs}
        return self._infer(audio, input_ids, labels, drops)

    def _infer(self, audio, input_ids, labels, drops):
        """No-grad path.  Eval-mode batches are processed in chunks (BatchNorm uses running statistics there,
        so chunking is exact); train-mode batches go through whole because the batch statistics need them."""
        ws = self._ws_infer
        B = audio.shape[0] if audio is not None and self.modality != "text" else input_ids.shape[0]
        ncls = self.classifier[-1].weight.shape[0]
        logits = torch.empty((B, ncls), dtype=torch.float32, device=self._arena.device)
        batch_stats = self.training or (self.audio_model is not None and audio_engine.bn_training(self.audio_model))
        chunk = B if batch_stats else INFER_CHUNK
        run = None
        for i0 in range(0, B, max(chunk, 1)):
            i1 = min(B, i0 + chunk)
            a = audio[i0:i1] if audio is not None and self.modality != "text" else None
            t = input_ids[i0:i1] if input_ids is not None and self.modality != "audio" else None
            run = self._run(a, t, None, drops, need_grad=False, ws=ws, logits_out=logits[i0:i1])
        if run is not None and run.text is not None:
            text_engine.check_ids(run.text.err)      # the flag is sticky across the chunks
        probs = torch.empty_like(logits)
        loss = None
        ce = torch.empty(1, dtype=torch.float32, device=logits.device) if labels is not None else None
        lab = None
        if labels is not None:
            lab = labels.contiguous().to(torch.int64)
            flag = torch.zeros(1, dtype=torch.int32, device=logits.device)
            ops.check_labels(lab, ncls, flag)
            text_engine.check_ids(flag)              # IndexError for a label outside the classes (synchronises)
        ops.softmax_ce(logits, lab, probs, ce, None)
        if labels is not None:
            loss = ce
            if self.modality == "both":
                psum = torch.empty(1, dtype=torch.float32, device=logits.device)
                ops.param_norms(self._arena.flat, self._arena.table, self._arena.tnorm, psum)
                loss = torch.empty_like(ce)
                ops.scalar_axpy(loss, ce, psum, L2_LAMBDA)
            loss = loss.view(())
        return {"loss": loss, "logits": logits, "probabilities": probs}


class FocalLoss(nn.Module):
    """multimodal_model.py:468-480.  The reference's drivers construct it but never call it
    (`multimodal_train.py:145-217` ignores ``criterion``), so it stays a plain composition of torch ops."""

    def __init__(self, alpha=None, gamma=2.0, reduction="mean"):
        super().__init__()
        self.alpha = alpha
        self.gamma = gamma
        self.reduction = reduction
        self.ce_loss = nn.CrossEntropyLoss(weight=alpha, reduction="none")

    def forward(self, inputs, targets):
        ce = self.ce_loss(inputs, targets)
        focal = (1 - torch.exp(-ce)) ** self.gamma * ce
        return focal.mean() if self.reduction == "mean" else focal.sum()

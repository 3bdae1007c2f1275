"""``torch.ops.mmsa.*``: the C ABI's launchers registered as torch custom operators (CUDA / HIP dispatch key only).

Every operator is a thin shim: it checks shapes (in :mod:`mmsa_amd.ops`) and enqueues one entry point of
``libmmsa_hip.so`` on torch's current stream.  Outputs are caller-allocated tensors marked as mutated in the schema
(the library allocates nothing), so the operators return nothing.  There is no CPU implementation: calling one with
CPU tensors fails in the dispatcher, which is the "no fallback" rule of this package in torch's own words.

Who calls what: the stand-alone ``forward`` of the drop-in sub-modules (``SEBlock``, ``ResidualBlock``,
``MultifactorFeatureNormalizer``, ``ModalityFusion``) and the batched front end go through these operators; the fused
training step (:class:`mmsa_amd.trainer.TrainStep`) and the whole-model autograd node call the same Python launchers
directly, because a captured or eagerly replayed sequence of ~100 launches per step should not pay a dispatcher
round trip per launch.  Both routes end in the same ``extern "C"`` symbols.

Offsets of a convolution's taps are passed as two int lists (``dh``, ``dw``), lists of device buffers as ``Tensor[]``.
"""
from __future__ import annotations

import torch

from . import ops

_LIB = torch.library.Library("mmsa", "DEF")
REGISTERED = []


def _reg(name: str, schema: str, fn):
    _LIB.define(name + schema)
    if schema.rstrip().endswith("-> ()"):
        launcher = fn

        def fn(*args, **kwargs):          # the launchers hand their output tensor back; the operator returns nothing
            launcher(*args, **kwargs)
    _LIB.impl(name, fn, "CUDA")
    REGISTERED.append(name)


def _taps(dh, dw):
    return list(zip(dh, dw))


# ---- front end (preprocess.py:127-134, 137-140, 151-168) ------------------------------------------------------------
_reg("logmel", "(Tensor wave, Tensor window, Tensor tw512, Tensor tw1024, Tensor melfb, Tensor melrange, "
               "Tensor(a!) out, int n_target, float top_db, Tensor? specmask) -> ()",
     lambda wave, window, tw512, tw1024, melfb, melrange, out, n_target, top_db, specmask:
     ops.logmel(wave, {"window": window, "tw512": tw512, "tw1024": tw1024, "melfb": melfb, "melrange": melrange}, out,
                n_target, top_db, specmask))
_reg("resample", "(Tensor pcm, Tensor kernT, int orig, int new, int width, int K, Tensor(a!) out) -> ()",
     lambda pcm, kernT, orig, new, width, K, out:
     ops.resample(pcm, {"kernT": kernT, "orig": orig, "new": new, "width": width, "K": K}, out))
_reg("augment_wave", "(Tensor wave, Tensor params, Tensor(a!) out, float noise_factor, int seed) -> ()",
     lambda wave, params, out, noise_factor, seed: ops.augment_wave(wave, params, out, noise_factor, seed))

# ---- convolutions (multimodal_model.py:48-59, 64-68, 152-155, 186-188) ---------------------------------------------
_reg("conv_pack_weights", "(Tensor w, Tensor(a!)? fwd, Tensor(b!)? dgrad) -> ()",
     lambda w, fwd, dgrad: ops.pack_weights(w, fwd, dgrad))
_reg("conv_fwd", "(Tensor X, Tensor Wp, Tensor(a!) Y, Tensor(b!)? stats, int B, int H, int W, int Cin, int Cout, "
                 "int[] dh, int[] dw, bool negate=False, bool accumulate=False) -> ()",
     lambda X, Wp, Y, stats, B, H, W, Cin, Cout, dh, dw, negate=False, accumulate=False:
     ops.conv_fwd(X, Wp, Y, stats, B, H, W, Cin, Cout, _taps(dh, dw), negate, accumulate))
_reg("conv_fwd_dual", "(Tensor X, Tensor Wp, Tensor X2, Tensor Wp2, Tensor(a!) Y, int B, int H, int W, int Cin, int Cout, "
                      "int[] dh, int[] dw, bool negate=False) -> ()",
     lambda X, Wp, X2, Wp2, Y, B, H, W, Cin, Cout, dh, dw, negate=False:
     ops.conv_fwd_dual(X, Wp, X2, Wp2, Y, B, H, W, Cin, Cout, _taps(dh, dw), negate))
_reg("conv_fwd_multi", "(Tensor X, Tensor[] Wps, Tensor(a!)[] Ys, int[] ntaps, int B, int H, int W, int Cin, int Cout, "
                       "int[] dh, int[] dw, int[] tap0, Tensor(b!)[] stats) -> ()",
     lambda X, Wps, Ys, ntaps, B, H, W, Cin, Cout, dh, dw, tap0, stats:
     ops.conv_fwd_multi(X, list(Wps), list(Ys), list(ntaps), B, H, W, Cin, Cout, _taps(dh, dw),
                        list(tap0) if len(tap0) else None, list(stats) if len(stats) else None))
_reg("conv_wgrad", "(Tensor X, Tensor dY, Tensor(a!) dW, Tensor(b!) workspace, int B, int H, int W, int Cin, int Cout, "
                   "int[] dh, int[] dw, bool accumulate=False) -> ()",
     lambda X, dY, dW, workspace, B, H, W, Cin, Cout, dh, dw, accumulate=False:
     ops.conv_wgrad(X, dY, dW, workspace, B, H, W, Cin, Cout, _taps(dh, dw), None, accumulate))
_reg("conv_wgrad_slabs", "(Tensor X, Tensor dY, Tensor? dY_sc, Tensor(a!) slabs, Tensor(b!)? slabs_sc, int B, int H, "
                         "int W, int Cin, int Cout) -> int",
     lambda X, dY, dY_sc, slabs, slabs_sc, B, H, W, Cin, Cout:
     ops.conv_wgrad_slabs(X, dY, dY_sc, slabs, slabs_sc, B, H, W, Cin, Cout))

# ---- audio branch, element-wise parts (multimodal_model.py:27-42, 64-70, 78-83, 118-127) ---------------------------
_reg("stem_stats", "(Tensor x, Tensor w, Tensor bias, Tensor(a!) partial, int B, int H, int W) -> ()", ops.stem_stats)
_reg("stem_apply_pool", "(Tensor x, Tensor w, Tensor bias, Tensor coeff, Tensor(a!) out, int B, int H, int W) -> ()",
     ops.stem_apply_pool)
_reg("bn_finalize", "(Tensor? partial, int rows, int tiles_per_img, int count, Tensor gamma, Tensor beta, "
                    "Tensor(a!)? running_mean, Tensor(b!)? running_var, Tensor(c!)? num_batches_tracked, bool training, "
                    "Tensor(d!) coeff, Tensor(e!)? bsum, int C) -> ()", ops.bn_finalize)
_reg("bn_relu_apply", "(Tensor x, Tensor coeff, Tensor(a!) y, int npos, int C) -> ()", ops.bn_relu_apply)
_reg("se_gate", "(Tensor bsum, Tensor coeff, Tensor W1, Tensor W2, int HW, Tensor(a!) gap, Tensor(b!) hid, "
                "Tensor(c!) gate, int B, int C) -> ()", ops.se_gate)
_reg("res_gate_relu_pool", "(Tensor c2, Tensor sc, Tensor coeff2, Tensor coeffs, Tensor gate, Tensor(a!) out, int B, "
                           "int H, int W, int C) -> ()", ops.res_gate_relu_pool)
_reg("gate_affine_add", "(Tensor a, Tensor? coeff2, Tensor? gate, Tensor? sc, Tensor? coeffs, Tensor(a!) out, int B, "
                        "int P, int C, bool relu) -> ()", ops.gate_affine_add)
_reg("gap_fwd", "(Tensor x, Tensor(a!) out, int B, int P, int C) -> ()", ops.gap_fwd)
_reg("gap_bwd", "(Tensor dout, Tensor(a!) dx, int B, int P, int C) -> ()", ops.gap_bwd)

# ---- text branch (multimodal_model.py:149, 179, 186-197) -----------------------------------------------------------
_reg("embedding_gather", "(Tensor ids, Tensor table, Tensor(a!) out, Tensor(b!)? err=None) -> ()", ops.embedding_gather)
_reg("bias_relu_maxtime_multi", "(Tensor[] ys, Tensor[] biases, int[] louts, Tensor(a!) feat, Tensor(b!) arg, int B, "
                                "int L, int F) -> ()",
     lambda ys, biases, louts, feat, arg, B, L, F:
     ops.bias_relu_maxtime_multi(list(ys), list(biases), list(louts), feat, arg, B, L, F))

# ---- head (multimodal_model.py:202-285, 365-404, 448-454) ----------------------------------------------------------
_reg("dense_fwd", "(Tensor X, Tensor W, Tensor? bias, Tensor? gamma, Tensor? beta, Tensor(a!) Y, Tensor(b!)? pre, "
                  "Tensor(c!)? stats, int epilogue=0, float drop_p=0.0, int seed=0, Tensor? step=None) -> ()", ops.dense_fwd)
_reg("ln_scale_fwd", "(Tensor x, Tensor gamma, Tensor beta, Tensor scale, Tensor? keep, Tensor(a!) y, Tensor(b!) stats)"
                     " -> ()", ops.ln_scale_fwd)
_reg("ln_scale_bwd", "(Tensor dy, Tensor x, Tensor gamma, Tensor beta, Tensor scale, Tensor? keep, Tensor stats, "
                     "Tensor(a!) dx, Tensor(b!) dgamma, Tensor(c!) dbeta, Tensor(d!) dscale, Tensor(e!) scratch) -> ()",
     ops.ln_scale_bwd)
_reg("fusion_mix_fwd", "(Tensor P, Tensor att, Tensor(a!) wts, Tensor(b!) fused) -> ()", ops.fusion_mix_fwd)
_reg("fusion_mix_bwd", "(Tensor dfused, Tensor P, Tensor wts, Tensor(a!) dP, Tensor(b!) datt) -> ()", ops.fusion_mix_bwd)
_reg("softmax_ce", "(Tensor logits, Tensor? labels, Tensor(a!) probs, Tensor(b!)? loss, Tensor(c!)? dlogits) -> ()",
     ops.softmax_ce)
_reg("dropout", "(Tensor x, Tensor(a!) y, float drop_p, int seed, Tensor? step=None) -> ()", ops.dropout)

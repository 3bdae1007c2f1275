"""GPU parity of the fused training step (forward + backward + L2 term + clip + AdamW) against the oracle's
written-out step and against the reference's own 100-step loss curve (tests/golden/loss_curves.json:
reference model + torch.optim.AdamW + clip_grad_norm_, B=32, lr 1e-3, wd 1e-4, dropout 0, seed 42)."""
import os

import numpy as np
import pytest
import torch

from conftest import SMALL_VOCAB, batch_tensors, model_inputs, seeded_model
from oracle import train_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _step_inputs(modality, b):
    kw = model_inputs(modality, b)
    kw["labels"] = b["label"]
    return kw


@pytest.mark.parametrize("modality", ["text", "audio", "both"])
def test_three_steps_match_oracle(modality):
    from mmsa_amd.trainer import TrainStep
    m = seeded_model(modality, vocab=SMALL_VOCAB).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    ref = train_ref.TrainState(sd0, modality)
    draws = [(False, False), (True, False), (False, True)]
    for s in range(3):
        b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB)
        if modality == "both":
            m.fusion_layer.draw_modality_dropout = lambda d=draws[s]: d
        loss = ts.step(**_step_inputs(modality, {k: v.to(DEV) for k, v in b.items()}))
        r = ref.step(b["audio"], b["input_ids"], b["label"], *(draws[s] if modality == "both" else (False, False)))
        assert abs(float(loss) - r["loss"]) < 1e-4, (s, float(loss), r["loss"])
        # the first step's norm agrees to rounding; from the second on the two runs no longer hold the same weights
        # (Adam's first update is +-lr per element, with the sign of gradients that are rounding noise), and the norm
        # of the gradient follows the weights: 1e-3 measured at the third step
        tol = 1e-5 if s == 0 else 2e-3
        assert abs(float(ts.grad_norm()) - r["grad_norm"]) < tol * max(1.0, r["grad_norm"])
    # After three Adam steps the accumulated UPDATE of every tensor must agree (relative L2).  Adam divides by
    # sqrt(v), so elements whose gradient is rounding noise move by +-lr with a sign that differs between any
    # two fp32 implementations; the L2 metric over the whole tensor is insensitive to those few elements.
    for k, p in m.named_parameters():
        a, bb, p0 = p.detach().cpu().double(), ref.sd[k].double(), sd0[k].double()
        if "conv1.0.bias" in k or (modality == "audio" and k in ("classifier.1.bias", "classifier.5.bias")):
            continue     # true gradient is exactly zero (a bias in front of a BatchNorm): pure noise
        upd = float(torch.linalg.vector_norm(bb - p0))
        err = float(torch.linalg.vector_norm(a - bb))
        assert err < 0.08 * upd + 1e-6, f"{k}: update error {err:.3e} vs update size {upd:.3e}"


@pytest.mark.parametrize("modality,B,L,W", [("both", 1, 128, 130), ("both", 5, 128, 130), ("both", 33, 128, 130),
                                            ("both", 64, 128, 130), ("audio", 3, 128, 130), ("text", 2, 128, 130),
                                            ("both", 4, 50, 91), ("both", 6, 17, 40), ("text", 3, 7, 130)])
def test_first_step_at_other_batch_sizes(modality, B, L, W):
    """A replayed step at batch sizes the kernels do not tile by (one utterance included), and at other token counts
    and spectrogram widths: loss and clip norm against the oracle."""
    from mmsa_amd.trainer import TrainStep
    m = seeded_model(modality, vocab=SMALL_VOCAB).to(DEV).train()
    sd0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    if modality == "both":
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0, use_graph=True)
    ref = train_ref.TrainState(sd0, modality)
    b = batch_tensors(100, B, vocab=SMALL_VOCAB)
    b["input_ids"] = b["input_ids"][:, :L].contiguous()
    b["audio"] = b["audio"][..., :W].contiguous()
    loss = ts.step(**_step_inputs(modality, {k: v.to(DEV) for k, v in b.items()}))
    r = ref.step(b["audio"], b["input_ids"], b["label"], False, False)
    assert abs(float(loss) - r["loss"]) < 1e-4, (float(loss), r["loss"])
    assert abs(float(ts.grad_norm()) - r["grad_norm"]) < 2e-5 * max(1.0, r["grad_norm"])
    ts.step(**_step_inputs(modality, {k: v.to(DEV) for k, v in b.items()}))      # the replay itself runs at this size
    assert bool(torch.isfinite(m._arena.flat).all())


def test_a_short_batch_in_graph_mode_is_launched_eagerly():
    """A batch of another size between replayed steps (a loader without drop_last): that step is launched kernel by
    kernel and the captured sequence keeps serving the usual shape; a validation pass runs in between; the run equals
    the all-eager run bit for bit.  (The descriptor tables a captured launch reads must survive the other shapes.)"""
    from mmsa_amd.trainer import TrainStep
    sizes = [8, 8, 8, 8, 5, 8, 3, 8]
    res = []
    for use_graph in (True, False):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0, use_graph=use_graph)
        losses = []
        for s, B in enumerate(sizes):
            b = batch_tensors(8 * s, B, vocab=SMALL_VOCAB)
            losses.append(ts.step(**_step_inputs("both", {k: v.to(DEV) for k, v in b.items()})).clone())
            if s == 3:      # a validation pass between training steps, at yet another batch size
                m.eval()
                v = batch_tensors(500, 11, vocab=SMALL_VOCAB)
                with torch.no_grad():
                    losses.append(m(**model_inputs("both", {k: t.to(DEV) for k, t in v.items()}),
                                    labels=v["label"].to(DEV))["loss"].clone())
                m.train()
        assert (ts._graph is not None) == use_graph
        res.append((torch.stack([x.reshape(()) for x in losses]), m._arena.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]), (res[0][0] - res[1][0]).abs()
    assert torch.equal(res[0][1], res[1][1])


def test_learning_rate_change_between_replayed_steps():
    """A scheduler's learning-rate change (multimodal_train.py:314-320) in graph mode: the step is captured again with
    the new rate; the run equals the eager run bit for bit, and the optimiser state round-trips through state_dict."""
    from mmsa_amd.trainer import TrainStep
    res = []
    for use_graph in (True, False):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0, use_graph=use_graph)
        losses = []
        for s in range(9):
            if s == 4:
                ts.set_lr(2.5e-4)
            if s == 7:
                ts.set_lr(2.5e-4)          # unchanged: no new capture
                g_before = ts._graph
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB)
            losses.append(ts.step(**_step_inputs("both", {k: v.to(DEV) for k, v in b.items()})).clone())
            if s == 7 and use_graph:
                assert ts._graph is g_before
        assert ts.state_dict()["param_groups"][0]["lr"] == 2.5e-4
        # the other scalars a captured step holds by value: assigning to them takes effect too
        ts.wd, ts.max_norm = 0.05, 0.5
        for s in range(9, 12):
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB)
            losses.append(ts.step(**_step_inputs("both", {k: v.to(DEV) for k, v in b.items()})).clone())
        assert float(ts.grad_norm()) > 0.5          # the clip threshold in force is the new one ...
        res.append((torch.stack([x.reshape(()) for x in losses]), m._arena.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]), (res[0][0] - res[1][0]).abs()
    assert torch.equal(res[0][1], res[1][1])         # ... and the replayed run equals the eager one with it


def test_reference_scheduler_drives_the_step_through_a_carrier_optimizer():
    """The reference builds ReduceLROnPlateau(optimizer, mode='min', factor=0.5, patience=2) on its AdamW and steps it
    once per epoch (multimodal_train.py:499-504, 317-318).  Schedulers accept stock optimisers only, so the drop-in
    keeps that AdamW as the object the scheduler drives and TrainStep(lr_from=...) follows its rate: same bits as
    setting the rates by hand, in graph mode."""
    from torch.optim.lr_scheduler import ReduceLROnPlateau
    from mmsa_amd.trainer import TrainStep
    res = []
    for follow in (True, False):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        if follow:
            carrier = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
            sched = ReduceLROnPlateau(carrier, mode="min", factor=0.5, patience=0)
            ts = TrainStep(m, weight_decay=1e-4, max_norm=1.0, use_graph=True, lr_from=carrier)
        else:
            ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0, use_graph=True)
        losses, rates = [], []
        for s in range(9):
            if s in (3, 6):                                  # an "epoch" ends; validation loss got worse each time
                if follow:
                    sched.step(1.0 + s)
                else:
                    ts.set_lr(1e-3 if s == 3 else 5e-4)      # patience 0: the first bad epoch after the best halves
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB)
            losses.append(ts.step(**_step_inputs("both", {k: v.to(DEV) for k, v in b.items()})).clone())
            rates.append(ts.lr)
        if follow:
            assert rates == [1e-3] * 6 + [5e-4] * 3, rates
            assert ts.state_dict()["param_groups"][0]["lr"] == carrier.param_groups[0]["lr"] == 5e-4
            assert not carrier.state_dict()["state"]          # the carrier never stepped
        res.append((torch.stack([x.reshape(()) for x in losses]), m._arena.flat.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


def test_train_step_refuses_a_partly_frozen_model():
    """torch.optim.AdamW skips parameters without a gradient; the fused step would decay and move them: it says so."""
    from mmsa_amd.trainer import TrainStep
    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    for p in m.text_model.parameters():
        p.requires_grad_(False)
    with pytest.raises(ValueError, match="frozen"):
        TrainStep(m)
    # the autograd path serves that case: frozen parameters receive no gradient
    m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    b = batch_tensors(0, 4, vocab=SMALL_VOCAB, device=DEV)
    m(**model_inputs("both", b), labels=b["label"])["loss"].backward()
    assert all(p.grad is None for p in m.text_model.parameters())
    assert all(p.grad is not None for p in m.audio_model.parameters())


def test_reference_amp_loop_on_the_drop_in_model():
    """The reference's --use_amp branch (multimodal_train.py:178-189: autocast around the forward, GradScaler around
    backward / unscale_ / clip / step) on the drop-in model: the kernels compute in float32 whatever the autocast state
    and the loss scale is a power of two, so the losses equal those of the plain loop."""
    res = []
    for use_amp in (False, True):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
        scaler = torch.amp.GradScaler("cuda") if use_amp else None
        losses = []
        for s in range(3):
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
            opt.zero_grad()
            if use_amp:
                with torch.autocast("cuda"):
                    loss = m(**model_inputs("both", b), labels=b["label"])["loss"]
                scaler.scale(loss).backward()
                scaler.unscale_(opt)
                torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
                scaler.step(opt)
                scaler.update()
            else:
                loss = m(**model_inputs("both", b), labels=b["label"])["loss"]
                loss.backward()
                torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=1.0)
                opt.step()
            losses.append(float(loss.detach()))
        res.append(losses)
    assert res[0] == res[1]


def _record(name, payload):
    """Measured deviations go to gpurun_out/ (merged back from the GPU box; the copies judged are in profiles/)."""
    import json
    from conftest import ROOT
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
            json.dump(payload, f)
    except OSError:
        pass


def test_loss_curve_100_steps_matches_reference(golden):
    """100 free-running training steps from seed 42 against the curve recorded from the reference (its model + torch
    AdamW + clip_grad_norm_ on CPU), with the recorded modality-dropout decisions reproduced by the host RNG.

    The first five steps agree to 2e-4.  Over 100 steps two fp32 runs of this network separate chaotically (Adam
    normalises noise-level gradients to +-lr, max-pool near-ties reroute gradients): the CPU oracle itself -- the same
    torch CPU ops as the reference, only the AdamW arithmetic written out by hand -- deviates from the recording by
    up to 1.3e-2 (mean 1.8e-3; series "oracle_cpu_fp32" of tests/golden/loss_curve_sensitivity.json), and by 2.6e-2
    (mean 4.0e-3) with oneDNN switched off.  The step is bit-reproducible, so the HIP curve is one fixed sample per
    build, but WHICH sample is decided by last-bit differences: eight builds of the kernels gave max 1.1e-2 .. 3.8e-2,
    mean 2.8e-3 .. 4.1e-3; taking the clip norm's embedding share from the batch's rows instead of the whole table
    (2e-7 of the norm) moved the maximum from 2.3e-2 to 3.8e-2, both at step 69, where the recorded curve has a spike.
    A bar of 2x the CPU series on a single sample would therefore pass or fail by the luck of the build.  What is
    asserted instead, besides the loose single-sample bars (max < 4x, mean < 3x the CPU series):

      the recorded reference curve must be a plausible member of the family of HIP curves.  Eight HIP runs start from
      the seed-42 weights perturbed by at most one unit in the last place (run 0: unperturbed); at every step the
      reference loss must lie within 4 standard deviations of the runs' mean (plus 3e-4 for the systematic 7e-5 .. 1.2e-4
      of the oracle's fp32 parameter norm, see test_teacher_forced_100_steps), with at most five exceptions in a
      hundred steps (with eight runs the deviation estimate follows a t-distribution: about one step in a hundred
      exceeds 4 sigma by chance).  A step that ran differently from the reference's -- another clip coefficient, a
      dropped regulariser, a wrong Adam bias correction -- moves the whole family away from the recording within a few
      steps, where the family is still tight (its standard deviation is 1e-4 at step 3, 1e-3 at step 10, 2.4e-3 at step
      50; measured: the recording leaves the 4-sigma band at one step of the hundred, and the eight members' own maximum
      deviations from it range from 1.0e-2 to 4.2e-2 -- two of the eight would fail a 2x bar).

    A per-step bound that is free of the accumulation is asserted by test_teacher_forced_100_steps below."""
    import json
    from conftest import GOLDEN
    from mmsa_amd.trainer import TrainStep
    ref = golden["loss_curves"]["both"]
    with open(os.path.join(GOLDEN, "loss_curve_sensitivity.json")) as f:
        cpu = np.array(json.load(f)["deviation"]["oracle_cpu_fp32"])
    want = np.array(ref["loss"])
    batches = [batch_tensors(32 * s, 32, device=DEV) for s in range(100)]
    curves = []
    for k in range(8):
        m = seeded_model("both").to(DEV).train()      # seed 42: the host RNG now sits where the reference's did
        ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0, use_graph=k > 0)
        if k > 0:
            g = torch.Generator(device=DEV).manual_seed(1000 + k)
            ulp = torch.randint(-1, 2, ts.arena.flat.shape, generator=g, device=DEV).float() * 2.0 ** -23
            ts.arena.flat.mul_(1.0 + ulp)
            # (run 0 takes its modality-dropout decisions from the host RNG, which must reproduce the recorded ones)
            m.fusion_layer.draw_modality_dropout = iter([tuple(bool(v) for v in d) for d in ref["drops"]]).__next__
        losses = [ts.step(**_step_inputs("both", b)).clone() for b in batches]
        curves.append(torch.stack(losses).flatten().cpu().numpy())
        if k == 0:
            m.eval()
            b = batch_tensors(100000, 32, device=DEV)
            with torch.no_grad():
                final_eval = float(m(**model_inputs("both", b), labels=b["label"])["loss"])
    curves = np.stack(curves)
    got = curves[0]
    err = np.abs(got - want)
    mean, std = curves.mean(axis=0), curves.std(axis=0, ddof=1)
    dev = np.abs(want - mean)
    tol = 4.0 * std * np.sqrt(1.0 + 1.0 / len(curves)) + 3e-4
    outside = np.nonzero(dev > tol)[0]
    _record("loss_curve_free_running.json",
            {"what": "|HIP loss - recorded reference loss| per step, free-running, B=32, seed 42, 100 steps; family = 8 "
                     "runs from weights perturbed by <= 1 ulp",
             "deviation": err.tolist(), "max": float(err.max()), "mean": float(err.mean()),
             "first5_max": float(err[:5].max()), "cpu_oracle_max": float(cpu.max()), "cpu_oracle_mean": float(cpu.mean()),
             "family_std": std.tolist(), "reference_minus_family_mean": (want - mean).tolist(),
             "steps_outside_4_sigma": outside.tolist(),
             "family_max_dev": [float(np.abs(c - want).max()) for c in curves],
             "family_mean_dev": [float(np.abs(c - want).mean()) for c in curves]})
    print(f"loss-curve deviation: first5 max {err[:5].max():.2e}, overall max {err.max():.2e} at step "
          f"{int(err.argmax())}, mean {err.mean():.2e}; CPU oracle vs recording: max {cpu.max():.2e} mean {cpu.mean():.2e}; "
          f"family of 8: max dev {[round(float(np.abs(c - want).max()), 4) for c in curves]}, std at steps 3/10/50/99 "
          f"{std[3]:.1e}/{std[10]:.1e}/{std[50]:.1e}/{std[99]:.1e}, reference outside 4 sigma at steps {outside.tolist()}")
    assert err[:5].max() < 2e-4
    assert len(outside) <= 5, outside
    assert err.max() < 4 * cpu.max(), f"max loss deviation {err.max():.2e} at step {int(err.argmax())}"
    assert err.mean() < 3 * cpu.mean()
    assert abs(got[-10:].mean() - want[-10:].mean()) < 2 * cpu.max()
    assert abs(final_eval - ref["final_eval_loss"]) < 4 * cpu.max()


def test_teacher_forced_100_steps(golden):
    """The reference's 100-step run, step by step without accumulation: before every step the HIP trainer is given
    the CPU oracle's parameters, BatchNorm buffers and Adam moments, both run the step on the same batch with the
    recorded modality-dropout decision, and are compared:
      * loss within 2e-4.  Measured: 0.7e-4 .. 1.2e-4, all of it one systematic term: the reference's loss contains
        0.001 * sum ||p||, and torch's fp32 CPU norm of the 7.8 M-element embedding table is 2.5e-4 low (279.391
        against 279.461 in float64), i.e. 7e-5 of loss, growing with the table's norm; the HIP kernel accumulates in
        double.  With the 1000-row table of test_three_steps_match_oracle the same comparison holds to 1e-5,
      * global gradient norm within 1e-4 relative,
      * the update p_after - p_before of every parameter tensor: relative L2 error against the oracle's update.
    The update bar is 1e-4 for the median tensor of a step (measured 3.0e-5) and 1.2e-2 for the worst tensor from
    step 10 on (measured 5.8e-3); in the first steps Adam's v is still ~g^2, so the update of an element is
    ~+-lr*sign(g) and elements whose gradient is rounding noise take different signs in any two fp32
    implementations (measured 2.7e-2, bounded there by 6e-2).  Tensors whose
    true gradient is identically zero (a bias in front of a BatchNorm) are pure noise and skipped."""
    from mmsa_amd.trainer import TrainStep
    ref = golden["loss_curves"]["both"]
    m = seeded_model("both").to(DEV).train()
    ts = TrainStep(m, lr=1e-3, weight_decay=1e-4, max_norm=1.0)
    st = train_ref.TrainState({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, "both")
    names = [k for k, _ in m.named_parameters()]
    assert names == st.keys
    skip = {"audio_model.conv1.0.bias"}
    rec = {"loss_err": [], "gnorm_rel_err": [], "update_median": [], "update_worst": [], "worst_tensor": [],
           "oracle_vs_recording": []}
    for s in range(100):
        b = batch_tensors(32 * s, 32)
        da, dt = ref["drops"][s]
        # the oracle's state BEFORE the step goes to the GPU trainer
        m.load_state_dict(st.sd, strict=True)
        if st.t > 0:
            ts.load_state_dict({"state": {i: {"step": torch.tensor(float(st.t)), "exp_avg": st.m[k], "exp_avg_sq": st.v[k]}
                                          for i, k in enumerate(st.keys)},
                                "param_groups": [{"lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 1e-4,
                                                  "params": list(range(len(st.keys)))}]})
        before = {k: st.sd[k].clone() for k in st.keys}
        r = st.step(b["audio"], b["input_ids"], b["label"], bool(da), bool(dt))
        m.fusion_layer.draw_modality_dropout = lambda d=(bool(da), bool(dt)): d
        loss = float(ts.step(**_step_inputs("both", {k: v.to(DEV) for k, v in b.items()})))
        gn = float(ts.grad_norm())
        rel = []
        for k, p in m.named_parameters():
            if k in skip:
                continue
            upd_ref = (st.sd[k] - before[k]).double()
            upd = (p.detach().cpu() - before[k]).double()
            rel.append((float(torch.linalg.vector_norm(upd - upd_ref) / (torch.linalg.vector_norm(upd_ref) + 1e-30)), k))
        vals = sorted(v for v, _ in rel)
        worst = max(rel)
        rec["loss_err"].append(abs(loss - r["loss"]))
        rec["gnorm_rel_err"].append(abs(gn - r["grad_norm"]) / max(r["grad_norm"], 1e-12))
        rec["update_median"].append(vals[len(vals) // 2])
        rec["update_worst"].append(worst[0])
        rec["worst_tensor"].append(worst[1])
        rec["oracle_vs_recording"].append(abs(r["loss"] - ref["loss"][s]))
    rec["what"] = ("teacher-forced: HIP step vs CPU oracle step from the oracle's state, 100 steps, B=32, seed 42; "
                   "update_* = relative L2 error of p_after - p_before per tensor")
    _record("loss_curve_teacher_forced.json", rec)
    a = {k: np.array(v) for k, v in rec.items() if k not in ("worst_tensor", "what")}
    print(f"teacher-forced: loss err max {a['loss_err'].max():.2e}; grad-norm rel err max {a['gnorm_rel_err'].max():.2e}; "
          f"update median max {a['update_median'].max():.2e}; worst tensor {a['update_worst'][:10].max():.2e} (steps 0-9) "
          f"{a['update_worst'][10:].max():.2e} (steps 10-99, {rec['worst_tensor'][10 + int(a['update_worst'][10:].argmax())]})")
    assert a["loss_err"].max() < 2e-4
    assert a["gnorm_rel_err"].max() < 1e-4
    assert a["update_median"].max() < 1e-4
    assert a["update_worst"][:10].max() < 6e-2
    assert a["update_worst"][10:].max() < 1.2e-2
    # every HIP step was within 2e-4 of an oracle step that started from the same state; the oracle's steps trace the
    # recorded reference curve (it free-runs here), so the recording is met to 1e-3 for as long as the oracle meets it
    # (how long depends on the host: the oracle's own rounding changes with the CPU's thread count -- 12 steps with 8
    # threads, 8 steps with 16)
    first = int(np.argmax(a["oracle_vs_recording"] > 9e-4)) if (a["oracle_vs_recording"] > 9e-4).any() else 100
    assert first >= 5


def test_graph_replay_equals_eager():
    """The captured HIP graph of the whole step must reproduce the eager launch sequence bit for bit, over all six
    steps (four of them replays)."""
    from mmsa_amd.trainer import TrainStep
    res = []
    for use_graph in (False, True):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        ts = TrainStep(m, use_graph=use_graph)
        ls = []
        for s in range(6):
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
            ls.append(ts.step(**_step_inputs("both", b)).clone())
        res.append((torch.stack(ls).cpu(), m._arena.flat.clone().cpu()))
    # same kernels, same order, and no sum in the step depends on the order in which workgroups run: bitwise
    assert torch.equal(res[0][0], res[1][0]), (res[0][0] - res[1][0]).abs().flatten()
    assert torch.equal(res[0][1], res[1][1])


def test_train_step_is_bit_reproducible():
    """Two runs of the metric's configuration (B=32, full vocabulary, the training driver's dropout defaults, graph
    replay, [CLS]/[SEP]/repeated word ids in every utterance) end in the same bits: losses, parameters, Adam moments."""
    from mmsa_amd.trainer import TrainStep
    res = []
    for _ in range(2):
        torch.manual_seed(42)
        m = seeded_model("both", dropout0=False).to(DEV).train()
        ts = TrainStep(m, use_graph=True)
        ls = []
        for s in range(8):
            b = batch_tensors(32 * s, 32, device=DEV)
            ls.append(ts.step(**_step_inputs("both", b)).clone())
        torch.cuda.synchronize()
        res.append((torch.stack(ls).cpu(), m._arena.flat.cpu().clone(), ts.m.cpu().clone(), ts.v.cpu().clone()))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)


@pytest.mark.parametrize("modality", ["both", "text"])
def test_clip_norm_from_the_batchs_embedding_rows_equals_the_dense_pass(modality, monkeypatch):
    """The clip norm (multimodal_train.py:194) reads the embedding table's gradient only at the rows of this batch's
    tokens -- everything else in it is zero, and the regulariser's share of that tensor's norm is lambda^2 whatever
    the rows hold -- and must agree with the pass over the whole table to rounding, step after step."""
    from mmsa_amd.trainer import TrainStep
    norms = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MMSA_SPARSE_NORM", mode)
        m = seeded_model(modality).to(DEV).train()
        if modality == "both":
            m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        ts = TrainStep(m)
        vals = []
        for s in range(3):
            b = batch_tensors(32 * s, 32, device=DEV)
            b["input_ids"][1, 7] = 30521              # the table's last row
            ts.step(**_step_inputs(modality, b))
            vals.append(float(ts.grad_norm()))
        norms[mode] = vals
    for a, c in zip(norms["1"], norms["0"]):
        assert abs(a - c) <= 2e-6 * c, (norms["1"], norms["0"])


def test_training_from_a_loader_with_worker_processes():
    """The reference's data path (multimodal_train.py:412-450: the dataset's transform runs in loader worker processes)
    with the split transform: workers return mono PCM (CPU only), the training process runs the GPU half on each
    collated batch and steps.  Two epochs over the same eight utterances in the same order give the same losses as
    transforming utterance by utterance in this process and stepping on the stacked maps."""
    from torch.utils.data import DataLoader, Dataset
    from mmsa_amd.frontend import AudioTransform, collate_utterances
    from mmsa_amd.trainer import TrainStep

    class Utterances(Dataset):
        def __init__(self, transform):
            self.transform = transform
            self.b = batch_tensors(0, 8, vocab=SMALL_VOCAB)

        def __len__(self):
            return 8

        def pcm(self, i):
            g = torch.Generator().manual_seed(50 + i)
            n = 30000 + 4000 * i                              # 16 kHz stereo utterances of different lengths
            t = torch.arange(n) / 16000.0
            tone = 0.2 * torch.sin(2 * np.pi * (200.0 + 40 * i) * t)
            return torch.stack([tone, tone]) + 0.01 * torch.randn(2, n, generator=g)

        def __getitem__(self, i):
            return {"audio": self.transform(self.pcm(i), 16000), "input_ids": self.b["input_ids"][i],
                    "attention_mask": (self.b["input_ids"][i] != 0).long(), "label": self.b["label"][i]}

    tf = AudioTransform(augment=False, device=DEV)
    ds = Utterances(tf.for_workers())
    loader = DataLoader(ds, batch_size=4, shuffle=False, num_workers=2, collate_fn=collate_utterances, drop_last=True)
    front = tf.batch_front_end()
    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    ts = TrainStep(m)
    got = []
    for _ in range(2):
        for batch in loader:
            audio = front(batch["audio"])                      # was: batch['audio'].to(DEVICE)
            assert audio.shape == (4, 1, 64, 130) and audio.is_cuda
            got.append(float(ts.step(audio=audio, input_ids=batch["input_ids"].to(DEV), labels=batch["label"].to(DEV))))
    # the same utterances through the per-utterance call of the reference's interface, in this process
    m2 = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    m2.fusion_layer.draw_modality_dropout = lambda: (False, False)
    ts2 = TrainStep(m2)
    strict = AudioTransform(augment=False, device=DEV, strict=True)
    maps = torch.stack([strict(ds.pcm(i), 16000) for i in range(8)]).to(DEV)
    want = []
    for _ in range(2):
        for lo in (0, 4):
            want.append(float(ts2.step(audio=maps[lo:lo + 4], input_ids=ds.b["input_ids"][lo:lo + 4].to(DEV),
                                       labels=ds.b["label"][lo:lo + 4].to(DEV))))
    assert len(got) == 4 and np.allclose(got, want, rtol=0, atol=2e-5), (got, want)


def test_train_from_waveforms_uses_frontend():
    from mmsa_amd.frontend import LogMelFrontEnd
    from mmsa_amd.trainer import TrainStep
    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    fe = LogMelFrontEnd(DEV)
    ts = TrainStep(m, frontend=fe)
    b = batch_tensors(0, 8, vocab=SMALL_VOCAB, device=DEV, wave=True)
    l1 = float(ts.step(wave=b["wave"], input_ids=b["input_ids"], labels=b["label"]))
    m2 = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    ts2 = TrainStep(m2)
    l2 = float(ts2.step(audio=fe(b["wave"]), input_ids=b["input_ids"], labels=b["label"]))
    assert l1 == l2 and np.isfinite(l1)


def test_autograd_path_with_torch_optimizer_matches_fused_step():
    """The reference's loop (zero_grad / forward / backward / clip_grad_norm_ / AdamW.step) run on the drop-in
    model gives the same trajectory as TrainStep."""
    from mmsa_amd.trainer import TrainStep
    torch.manual_seed(42)
    ma = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    mb = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    ma.fusion_layer.draw_modality_dropout = lambda: (False, False)
    mb.fusion_layer.draw_modality_dropout = lambda: (False, False)
    opt = torch.optim.AdamW(ma.parameters(), lr=1e-3, weight_decay=1e-4)
    ts = TrainStep(mb)
    for s in range(3):
        b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
        opt.zero_grad()
        out = ma(**model_inputs("both", b), labels=b["label"])
        out["loss"].backward()
        torch.nn.utils.clip_grad_norm_(ma.parameters(), max_norm=1.0)
        opt.step()
        lb = ts.step(**_step_inputs("both", b))
        assert abs(float(out["loss"].detach()) - float(lb)) < 1e-4
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        if "conv1.0.bias" in k:
            continue
        d = float(torch.linalg.vector_norm(pa.detach() - pb.detach()))
        assert d < 0.05 * 3e-3 * pa.numel() ** 0.5 + 1e-6, k      # a small fraction of three lr-sized steps


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_single_rank_process_group_path_equals_plain_step(overlap, monkeypatch):
    """With a process group (RCCL, one rank) the step runs as several graphs with the gradient exchange issued
    between them -- overlapped with the audio backward when overlap is on; the trajectory must equal the
    single-graph step's."""
    import torch.distributed as dist
    from mmsa_amd.trainer import TrainStep
    monkeypatch.setenv("MMSA_OVERLAP_ALLREDUCE", overlap)

    def run(ts_kwargs):
        m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        ts = TrainStep(m, **ts_kwargs)
        ls = []
        for s in range(6):
            b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
            ls.append(ts.step(**_step_inputs("both", b)).clone())
        torch.cuda.synchronize()
        return ts, torch.stack(ls).cpu(), m._arena.flat.clone().cpu()

    _, l_plain, p_plain = run(dict(use_graph=True))
    port = 29600 + (os.getpid() % 200)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        for use_graph in (True, False):
            ts, l_dist, p_dist = run(dict(use_graph=use_graph))
            assert ts.dist and ts.overlap == (overlap == "1")
            # the same kernels in the same order, cut into graphs at other places, and a one-rank sum: bitwise
            assert torch.equal(l_plain, l_dist), (l_plain - l_dist).abs().flatten()
            assert torch.equal(p_plain, p_dist)
    finally:
        dist.destroy_process_group()


def test_drop_in_model_inside_distributed_data_parallel():
    """The reference-style alternative to TrainStep's own data-parallel mode: the drop-in model wrapped in torch's
    DistributedDataParallel, whose reducer hangs hooks on every parameter's gradient accumulator and takes the gradients
    as the one autograd node of the fused model hands them over.  One rank over RCCL: losses and parameters equal the
    unwrapped run bit for bit."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    port = 29800 + (os.getpid() % 200)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        res = []
        for wrap in (False, True):
            m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
            m.fusion_layer.draw_modality_dropout = lambda: (False, False)
            net = DistributedDataParallel(m, device_ids=[0]) if wrap else m
            opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
            losses = []
            for s in range(4):
                b = batch_tensors(8 * s, 8, vocab=SMALL_VOCAB, device=DEV)
                opt.zero_grad()
                out = net(audio=b["audio"], input_ids=b["input_ids"], labels=b["label"])
                out["loss"].backward()
                torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
                opt.step()
                losses.append(float(out["loss"].detach()))
            res.append((losses, m._arena.flat.clone()))
        assert res[0][0] == res[1][0]
        assert torch.equal(res[0][1], res[1][1])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("modality,act,dact", [("both", "head.h1", "head.c0.dln"), ("text", "head.h1", "head.t1.dln"),
                                               ("audio", "head.y1", None)])
def test_graph_replay_draws_new_dropout_masks(modality, act, dact):
    """With the reference's default dropouts (multimodal_train.py:71-76; audio 0.5, text 0.3, fusion 0.5) the captured
    step works for every modality: the kernels mix the device-resident step count into the dropout seed, so
    consecutive replays drop different units, forward and backward of one step agree on the mask (a unit is zero in
    the forward activation exactly where its gradient is zero), and two runs draw the same masks."""
    from mmsa_amd.trainer import TrainStep
    runs = []
    for _ in range(2):
        m = seeded_model(modality, vocab=SMALL_VOCAB, dropout0=False).to(DEV).train()
        if modality == "both":
            m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        ts = TrainStep(m, use_graph=True)
        b = batch_tensors(0, 8, vocab=SMALL_VOCAB, device=DEV)
        width = {"both": 128, "text": 256, "audio": 128}[modality]
        zero_sets, losses = [], []
        for s in range(6):
            losses.append(float(ts.step(**_step_inputs(modality, b))))
            h1 = ts.ws.buf(act, (8, width))
            torch.cuda.synchronize()
            if dact is not None:
                assert torch.equal(h1 == 0, ts.ws.buf(dact, (8, width)) == 0)
            zero_sets.append((h1 == 0).cpu())
        assert ts._graph is not None and all(np.isfinite(losses))
        for a, c in zip(zero_sets[3:], zero_sets[4:]):           # replayed steps
            assert not torch.equal(a, c)
        dropped = float(zero_sets[-1].float().mean())
        lo = {"both": 0.55, "text": 0.25, "audio": 0.55}[modality]      # both / audio: relu zeros plus p = 0.5 dropout
        assert lo < dropped < 0.95, dropped
        runs.append((losses, m._arena.flat.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])


def test_token_id_outside_the_table_raises_and_touches_no_neighbouring_gradient():
    """The reference's nn.Embedding raises IndexError for a token id outside the table (multimodal_model.py:179).
    Here the forward substitutes a zero row and the backward drops that token's gradient, both raising a sticky
    device flag that the host turns into the same IndexError; nothing may be written outside the table's gradient."""
    from mmsa_amd import text as text_engine
    from mmsa_amd.model import TextModel
    from mmsa_amd.trainer import TrainStep
    from mmsa_amd.workspace import Grads, Workspace
    torch.manual_seed(0)
    V, E = 200, 256
    tm = TextModel(vocab_size=V).to(DEV)
    ids = torch.randint(1, V, (4, 128), device=DEV)
    ids[0, 3], ids[1, 7], ids[2, 0], ids[3, 127] = V, V + 40, -1, -(2 ** 40)
    ws = Workspace(DEV, persistent=True)
    feat, st = text_engine.text_forward(tm, ids, True, ws)
    assert int(st.err.item()) == 1
    st.err.zero_()
    guard = torch.full(((V + 128) * E,), 7.0, device=DEV)
    dtable = guard[64 * E:(64 + V) * E].view(V, E)
    dtable.zero_()
    grads = Grads({id(tm.embedding.weight): dtable})
    text_engine.text_backward(tm, st, torch.randn(4, 384, device=DEV), grads, ws)
    assert bool((guard[:64 * E] == 7.0).all()) and bool((guard[(64 + V) * E:] == 7.0).all())
    assert bool(torch.isfinite(dtable).all())
    with pytest.raises(IndexError):
        text_engine.check_ids(st.err)
    text_engine.check_ids(st.err)                 # the flag was cleared by the raise
    # drop-in model, eager forward: raises like the reference
    m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
    b = batch_tensors(0, 8, vocab=SMALL_VOCAB, device=DEV)
    bad = b["input_ids"].clone()
    bad[2, 5] = SMALL_VOCAB
    with pytest.raises(IndexError):
        m(audio=b["audio"], input_ids=bad, labels=b["label"])
    # fused step: host-resident ids are refused before anything is launched, device-resident ids at the next check
    ts = TrainStep(m)
    with pytest.raises(IndexError):
        ts.step(audio=b["audio"], input_ids=bad.cpu(), labels=b["label"])
    before = m._arena.flat.clone()
    ts.step(audio=b["audio"], input_ids=bad, labels=b["label"])
    with pytest.raises(IndexError):
        ts.check_ids()
    assert bool(torch.isfinite(m._arena.flat).all()) and not torch.equal(before, m._arena.flat)
    # without anybody asking: the flag travels to pinned host memory behind every step, and the first step() that
    # finds it there raises (here: the next one, because the GPU has finished); so does writing a checkpoint
    for use_graph in (False, True):
        ts = TrainStep(m, use_graph=use_graph)
        good = b["input_ids"]
        for _ in range(4):
            ts.step(audio=b["audio"], input_ids=good, labels=b["label"])
        ts.step(audio=b["audio"], input_ids=bad, labels=b["label"])
        with pytest.raises(IndexError):
            ts.state_dict()
        ts.step(audio=b["audio"], input_ids=bad, labels=b["label"])
        torch.cuda.synchronize()
        with pytest.raises(IndexError):
            ts.step(audio=b["audio"], input_ids=good, labels=b["label"])
        ts.step(audio=b["audio"], input_ids=good, labels=b["label"])        # the flag was cleared by the raise
        ts.state_dict()


def test_train_step_refuses_a_model_whose_parameters_moved_away():
    """TrainStep updates the flat storage the parameters lived in when it was built.  A model moved through host
    memory, or with an exchanged layer, has its parameters elsewhere: the step would train numbers nobody reads, so it
    refuses -- before and after a forward pass of the model has re-homed them -- and a new TrainStep takes over the
    optimiser state."""
    from mmsa_amd.trainer import TrainStep
    b = batch_tensors(0, 8, vocab=SMALL_VOCAB, device=DEV)
    for how in ("moved", "moved, then used", "layer exchanged"):
        for use_graph in (False, True):
            m = seeded_model("both", vocab=SMALL_VOCAB).to(DEV).train()
            ts = TrainStep(m, use_graph=use_graph)
            for _ in range(4):
                ts.step(**_step_inputs("both", b))
            if how == "layer exchanged":
                m.classifier[len(m.classifier) - 1] = torch.nn.Linear(64, 3).to(DEV)
            else:
                m.cpu().to(DEV)
                if how == "moved, then used":
                    with torch.no_grad():
                        m(audio=b["audio"], input_ids=b["input_ids"])
            with pytest.raises(RuntimeError, match="build a new TrainStep"):
                for _ in range(TrainStep.CHECK_EVERY + 1):       # a replayed step may take until the next full check
                    ts.step(**_step_inputs("both", b))
            if how != "layer exchanged":
                sd = ts.state_dict()
                ts2 = TrainStep(m, use_graph=use_graph)
                ts2.load_state_dict(sd)
                before = m._arena.flat.clone()
                ts2.step(**_step_inputs("both", b))
                assert not torch.equal(before, m._arena.flat)


@pytest.mark.parametrize("modality", ["audio", "text", "both"])
def test_class_label_outside_the_classes_raises(modality):
    """nn.CrossEntropyLoss refuses a target outside [0, C) (multimodal_model.py:452-454; the classic slip in this task
    is to pass sentiments as -1 / 0 / 1).  The loss kernels would treat such a row as one without a target: a small
    launch at the head of the pass raises a flag instead, which becomes the reference's IndexError -- at once in the
    eager forward (with or without gradients), at the next check in TrainStep."""
    from mmsa_amd.trainer import TrainStep
    m = seeded_model(modality, vocab=SMALL_VOCAB).to(DEV).train()
    b = batch_tensors(0, 8, vocab=SMALL_VOCAB, device=DEV)
    kw = model_inputs(modality, b)
    for bad_value in (-1, 3):
        bad = b["label"].clone()
        bad[5] = bad_value
        with pytest.raises(IndexError, match="out of bounds"):
            m(**kw, labels=bad)
        with pytest.raises(IndexError, match="out of bounds"):
            m(**kw, labels=bad.cpu())
        with pytest.raises(IndexError, match="out of bounds"), torch.no_grad():
            m(**kw, labels=bad)
    out = m(**kw, labels=b["label"])                          # the flag does not outlive the call that raised
    assert bool(torch.isfinite(out["loss"]))
    for use_graph in (False, True):
        ts = TrainStep(m, use_graph=use_graph)
        for _ in range(4):
            ts.step(**_step_inputs(modality, b))
        with pytest.raises(IndexError, match="out of bounds"):
            ts.step(**{**_step_inputs(modality, b), "labels": bad.cpu()})      # host-resident: before any launch
        ts.step(**{**_step_inputs(modality, b), "labels": bad})
        with pytest.raises(IndexError, match="out of bounds"):
            ts.state_dict()
        ts.step(**{**_step_inputs(modality, b), "labels": bad})
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match="out of bounds"):
            ts.step(**_step_inputs(modality, b))                                 # the pinned copy has landed
        ts.step(**_step_inputs(modality, b))
        ts.state_dict()

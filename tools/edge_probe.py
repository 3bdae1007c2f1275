#!/usr/bin/env python3
"""Ad-hoc probe: gradient parity of the drop-in model against the float64 oracle for unusual batch / sequence sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), ROOT, os.path.join(ROOT, "multimodal-sentiment-analysis_amd")):
    sys.path.insert(0, p)
import torch
from conftest import SMALL_VOCAB, seeded_model
from oracle import model_ref

def run(B, L, W, modality="both", seed=0):
    m = seeded_model(modality, vocab=SMALL_VOCAB).to("cuda").train()
    if modality == "both":
        m.fusion_layer.draw_modality_dropout = lambda: (False, False)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B * 7 + L + 1000 * seed)
    audio = torch.randn(B, 1, 64, W, generator=g)
    ids = torch.randint(1, SMALL_VOCAB, (B, L), generator=g)
    ids[:, L // 2:] = 0 if B > 1 else ids[:, L // 2:]
    ids[0, :] = 0 if B > 2 else ids[0, :]           # an all-padding utterance
    label = torch.randint(0, 3, (B,), generator=g)
    kw = {}
    if modality != "text": kw["audio"] = audio.cuda()
    if modality != "audio": kw["input_ids"] = ids.cuda()
    out = m(**kw, labels=label.cuda())
    out["loss"].backward()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64, g64, _ = model_ref.forward_backward(sd64, modality, audio.double() if modality != "text" else None,
                                             ids if modality != "audio" else None, label)
    sd32 = {k: v.clone() for k, v in sd.items()}
    _, g32, _ = model_ref.forward_backward(sd32, modality, audio if modality != "text" else None,
                                           ids if modality != "audio" else None, label)
    worst = ("", 0.0)
    rows = []
    for k, p in m.named_parameters():
        r = g64[k]
        if float(r.abs().max()) < 1e-6 or p.grad is None: continue
        nr = torch.linalg.vector_norm(r) + 1e-30
        err = float(torch.linalg.vector_norm(p.grad.detach().cpu().double() - r) / nr)
        e32 = float(torch.linalg.vector_norm(g32[k].double() - r) / nr)
        rows.append((err, e32, float(nr), k))
        if err > worst[1]: worst = (k, err)
    if os.environ.get("TOP"):
        for err, e32, nr, k in sorted(rows, reverse=True)[:int(os.environ["TOP"])]:
            print(f"    {k}: hip {err:.2e}  cpu-fp32 {e32:.2e}  |g| {nr:.2e}")
    dl = abs(float(out["loss"].detach()) - float(o64["loss"]))
    print(f"B={B} L={L} W={W} {modality}: loss diff {dl:.2e}, worst grad rel-L2 {worst[1]:.2e} ({worst[0]})", flush=True)

for B, L, W, mod, seed in [(b, 128, 130, "both", sd) for b in (16, 33) for sd in range(5)]:
    try:
        run(B, L, W, mod, seed)
    except Exception as e:
        print(f"B={B} L={L} W={W} {mod}: EXCEPTION {type(e).__name__}: {str(e)[:300]}", flush=True)

"""Developer aid (not a pytest file): per-tensor gradient error of the HIP path against the oracle."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import SMALL_VOCAB, batch_tensors, model_inputs, seeded_model  # noqa: E402
from oracle import model_ref  # noqa: E402

DEV = "cuda"


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for modality in ("text", "audio", "both"):
        m = seeded_model(modality, vocab=SMALL_VOCAB).to(DEV).train()
        if modality == "both":
            m.fusion_layer.draw_modality_dropout = lambda: (False, False)
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        b = batch_tensors(40, B, vocab=SMALL_VOCAB)
        bd = {k: v.to(DEV) for k, v in b.items()}
        out = m(**model_inputs(modality, bd), labels=bd["label"])
        out["loss"].backward()
        ref_out, ref_g, _ = model_ref.forward_backward(sd, modality, b["audio"], b["input_ids"], b["label"])
        # float64 oracle: how far is the fp32 CPU oracle itself from exact arithmetic?
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        o64, g64, _ = model_ref.forward_backward(sd64, modality, b["audio"].double(), b["input_ids"], b["label"])
        print(f"== {modality} B={B}: loss hip {float(out['loss']):.7f} cpu32 {float(ref_out['loss']):.7f} "
              f"cpu64 {float(o64['loss']):.7f}")
        le = float((out["logits"].detach().cpu().double() - o64["logits"]).abs().max())
        lc = float((ref_out["logits"].double() - o64["logits"]).abs().max())
        print(f"   logits max err vs f64: hip {le:.2e}  cpu32 {lc:.2e}")
        for k, p in m.named_parameters():
            g = p.grad.detach().cpu().double()
            r64 = g64[k]
            scale = float(r64.abs().max()) + 1e-30
            e_hip = float((g - r64).abs().max()) / scale
            e_cpu = float((ref_g[k].double() - r64).abs().max()) / scale
            flag = " <<<" if e_hip > 20 * e_cpu + 1e-5 else ""
            print(f"   {k:55s} scale {scale:.2e} hip {e_hip:.2e} cpu32 {e_cpu:.2e}{flag}")


if __name__ == "__main__":
    main()
